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

PARITY UNPINNED.  The reference ships no tests, no golden vectors and no
recorded outputs (all three ``results/`` directories are git-ignored), and its
path cannot be executed here: ``jax``, ``jaxlib``, ``osqp`` and ``ipyopt`` are
unpinned third-party dependencies (``requirements.txt:1-7``) that are not
installed in this image and cannot be fetched (no network).  The oracle is
therefore pinned only by (i) independent forward-mode autodiff
(``torch.func.jacfwd`` in fp64) over a separate restatement of the forward
functions, (ii) central finite differences, (iii) the structural invariants
the reference's code implies (causality, axis decoupling, ego
sample-independence, baseline == zero-noise special case) and (iv) analytic
CVaR identities — see ``tests/test_oracle_*.py``.  The golden fixtures under
``tests/golden/`` were produced by this oracle (``tests/golden/make_golden.py``).
"""
