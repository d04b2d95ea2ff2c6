"""Driver entry points: build() compiles the HIP library for gfx950; smoke() runs one tiny CUT step on cuda:0 and checks
it against the CPU oracle."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
CSRC = os.path.join(ROOT, "gan-variant-research_amd", "csrc")


def build() -> None:
    """hipcc --offload-arch=gfx950 for every kernel file -> gan-variant-research_amd/libmi355x_gan.so (in-tree), then
    import the package and check the exported ABI.  The oracle is Python (PyTorch CPU): nothing to compile there, and
    the reference is Python too, so there is no oracle/_ref build."""
    subprocess.run(["make", "-C", CSRC, "-j8"], check=True)
    import gan_variant_research_amd as pkg
    from gan_variant_research_amd import _lib
    lib = _lib.load()
    assert lib.gan_version() >= 100
    print(f"built {_lib.LIB_PATH}; {len(_lib.PROTOTYPES)} entry points bound")


def smoke() -> None:
    """One small CUT train step (B=2, 32x32, fp32 parity mode) on cuda:0 through the C ABI, checked against the oracle."""
    import torch
    from gan_variant_research_amd.runtime import HipOps
    from tests import cases
    assert torch.cuda.is_available(), "smoke() needs a GPU"
    tr, img, ref = cases.run_cut_steps("cuda:0", HipOps(torch.device("cuda:0")), True, amp=False, S=32, B=2, nsteps=1, tol0=1e-3)
    err = (img - ref).abs().max().item()
    assert err < 2e-3, err
    print(f"smoke ok: CUT step 0 losses match the CPU oracle within 1e-3; max |G(x) - oracle| = {err:.2e}")


if __name__ == "__main__":
    build()
    if len(sys.argv) > 1 and sys.argv[1] == "smoke":
        smoke()
