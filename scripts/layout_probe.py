"""Does the WALK of the backward sweep (gar_stream_sweep: its bytes, its waves, no arithmetic) run faster when the
records are laid out stage-major ([stage][problem]) instead of problem-major?  GAR_STREAM_LAYOUT=stage."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aligator_amd import _lib
L = _lib.load()
nx, nu, N = 36, 12, 256
knot_b = 8 * (nx * nx + nx * nu + nx * (nx + 1) // 2 + nx * nu + nu * (nu + 1) // 2 + 2 * nx + nu)
fac_b = 8 * ((nu + nx) * (nx + 1) + nx * (nx + 1) // 2 + nx)
for batch in (1024, 2048, 4096):
    for lay in ("problem", "stage"):
        os.environ["GAR_STREAM_LAYOUT"] = lay
        ms = L.gar_hip_stream_ceiling_ms(0, batch, N, knot_b, fac_b, 3)
        gb = (knot_b + fac_b) * batch * N / 1e9
        print(f"batch {batch:5d} layout {lay:8s}: walk {ms:7.3f} ms  {gb/ms:6.2f} TB/s (x1e-3)", flush=True)
