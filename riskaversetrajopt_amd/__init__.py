"""MI355X-native SAA inner loop for risk-averse trajectory optimization.

Drop-in for the hot path of StanfordASL/RiskAverseTrajOpt (batched rollout,
control-Jacobian linearization, sample mean, Monte-Carlo VaR/CVaR) behind the
reference's own ``Model`` constraint-evaluation API.  The compute runs in
hand-written HIP kernels (``csrc/``) reached through the C ABI of
``include/rato_saa.h``; PyTorch-ROCm only provides device memory and streams.
"""
__version__ = "0.1.0"

# Kernel arguments in device memory: the default of this ROCm stack, worth ~3 us per launch (an SCP round trip is four
# dependent launches: profiles/EXPERIMENTS.md section 6).  Stated here so that a stack with another default behaves the same;
# it only takes effect when set before the HIP runtime initialises (first device call), and never overrides the caller's choice.
import os as _os
_os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
