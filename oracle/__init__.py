"""CPU oracle for the SAA inner loop of StanfordASL/RiskAverseTrajOpt.

TEST INFRASTRUCTURE ONLY.  Nothing under ``riskaversetrajopt_amd/`` may import
this package; only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` do, and there only as the checker /
reported baseline, never as the thing shipped or measured as the product.

What it is: a NumPy fp64 restatement (vectorised over the sample axis) of the
reference's arithmetic for the hot path named in BASELINE.json — batched
rollout, control-Jacobian linearization, sample mean, Monte-Carlo constraint
check and VaR/CVaR — for the drone, driving and hopper problems.  Every
function cites the reference file:line it follows.

HOW PARITY IS PINNED.  The reference ships no tests, no golden vectors and no
recorded outputs (all three ``results/`` directories are git-ignored), and its
scripts cannot be imported as they stand: ``jax``, ``jaxlib``, ``osqp`` and
``ipyopt`` are unpinned third-party dependencies (``requirements.txt:1-7``) that
are not installed in this image and cannot be fetched (no network), and every
script runs its whole experiment at import time.  What does run here is the
reference's own ARITHMETIC TEXT: ``tests/golden/make_reference_golden.py``
extracts ``class Model`` (and the Monte-Carlo closures) from the files under
/root/reference with ``ast`` at run time and executes them unmodified against
``tests/golden/jax_standin.py`` (torch fp64 behind ``jnp`` / ``vmap`` /
``jacfwd`` / ``jacrev`` / ``hessian``), on samples drawn by the reference's own
sampler under its own seed; inputs and outputs are committed as
``tests/golden/ref_*.npz`` and ``tests/test_reference_pin.py`` asserts that this
oracle reproduces them to ~1e-11.  Caveat: jax itself is a stand-in (autodiff
comes from torch.func).  Independently of that the oracle is checked by (i)
forward-mode autodiff over a separate restatement of the forward functions,
(ii) central finite differences, (iii) the structural invariants the
reference's code implies (causality, axis decoupling, ego sample-independence,
baseline == zero-noise special case) and (iv) analytic CVaR identities — see
``tests/test_oracle_*.py``.  The fixtures ``tests/golden/{drone,driving,hopper}_*.npz``
were produced by this oracle (``tests/golden/make_golden.py``) as regression
vectors.
"""
