"""Import the reference's Python (read-only, /root/reference) on CPU.  THIS CONTAINER ONLY.

Used to (1) validate the oracle's restatement and (2) emit golden vectors under tests/golden/.
Nothing here ships reference code: stand-in modules for absent third-party packages live in
./stubs, the CUDA extension modules are replaced by the oracle's C restatement, and hard-coded
`.cuda()` calls are shimmed to CPU (so3net.py:21,49,151; pointtransformer_seg.py:59; pointops.py:21-22).
"""
import contextlib
import io
import os
import sys
import types

REF = "/root/reference"
_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(os.path.dirname(_HERE))


def available():
    return os.path.isdir(os.path.join(REF, "src", "models"))


_done = False


def setup():
    """Idempotently prepare sys.path + CPU shims.  Returns nothing."""
    global _done
    if _done:
        return
    assert available(), "reference tree not present"
    import torch

    for p in (os.path.join(REF, "src"), os.path.join(REF, "external", "vgtk"), os.path.join(_HERE, "stubs"), _ROOT):
        if p not in sys.path:
            sys.path.insert(0, p)
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.cuda.IntTensor = torch.IntTensor
    torch.cuda.FloatTensor = torch.FloatTensor
    _to = torch.nn.Module.to

    def to(self, *a, **k):
        def fix(x):
            if isinstance(x, torch.device) and x.type == "cuda":
                return torch.device("cpu")
            if isinstance(x, str) and x.startswith("cuda"):
                return "cpu"
            return x

        return _to(self, *[fix(x) for x in a], **{kk: fix(v) for kk, v in k.items()})

    torch.nn.Module.to = to
    _done = True


@contextlib.contextmanager
def quiet():
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        yield buf


def build_reference_model(tmpdir, markerset, radius=0.4, layers=2):
    """Construct the reference GT_network_equiv on CPU (random init; caller loads weights)."""
    setup()
    import torch

    with quiet():
        from models.models_pointcloud import GT_network_equiv
    opt = types.SimpleNamespace(
        output_folder=tmpdir, EPN_input_radius=radius, EPN_layer_num=layers, device=torch.device("cpu"), markerset=markerset
    )
    with quiet():
        m = GT_network_equiv(option=opt).eval()
    return m
