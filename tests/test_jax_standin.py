"""CPU: the stand-in that carries the reference's text when the golden fixtures are generated (tests/golden/jax_standin.py)
is itself checked -- every operation it maps against NumPy's semantics, and, where /root/reference exists, the Jacobians
the reference's own get_all_constraints_coeffs returns through it against central finite differences of the reference's
own forward text.  In a subprocess: installing the stand-in patches torch.Tensor."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))


def test_standin_semantics_and_reference_jacobians():
    out = subprocess.run([sys.executable, os.path.join(HERE, "golden", "check_standin.py")], capture_output=True, text=True,
                         timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert "standin ops ok" in out.stdout
    if os.path.isdir(os.environ.get("RATO_REFERENCE", "/root/reference")):
        assert "reference text ok" in out.stdout, out.stdout
