"""MI355X-native SAA inner loop for risk-averse trajectory optimization.

Drop-in for the hot path of StanfordASL/RiskAverseTrajOpt (batched rollout,
control-Jacobian linearization, sample mean, Monte-Carlo VaR/CVaR) behind the
reference's own ``Model`` constraint-evaluation API.  The compute runs in
hand-written HIP kernels (``csrc/``) reached through the C ABI of
``include/rato_saa.h``; PyTorch-ROCm only provides device memory and streams.
"""
__version__ = "0.1.0"
