"""Build-container-only check (skipped on the GPU box, where /root/reference does not exist): the
reference's OWN model package imports, constructs and loads a state dict on the MinkowskiEngine shim
-- i.e. the shim covers the operator surface model/*.py touches at import and construction time
(SURVEY 8b "Module import side effects").  Runs in a subprocess so the reference's `model` / `utils`
packages never shadow anything in this process."""
import os
import subprocess
import sys

import pytest

REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r"""
import torch, model
from model import load_model, fc
m = load_model("ResUNetBN2C")(1, 16, bn_momentum=0.05, normalize_feature=True, conv1_kernel_size=3, D=3)
e = fc.conv1_max_embedding(1024, 512, 256)
import sys
sys.path.insert(0, %r)
from corsair_amd import synth
sd, emb = synth.make_state_dicts(31)
assert set(sd) == set(m.state_dict()) and len(sd) == 129
assert all(tuple(m.state_dict()[k].shape) == tuple(sd[k].shape) for k in sd)
assert set(emb) == set(e.state_dict())
m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
e.load_state_dict({k: torch.from_numpy(v) for k, v in emb.items()})
assert sum(v.numel() for v in m.state_dict().values()) == 8753669   # SURVEY A.2
print("OK")
""" % ROOT


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "model")), reason="reference tree not present")
def test_reference_model_package_constructs_on_shim():
    env = dict(os.environ)
    env["PYTHONPATH"] = os.pathsep.join([os.path.join(ROOT, "shim"), ROOT, REF])
    out = subprocess.run([sys.executable, "-c", SCRIPT], cwd="/tmp", env=env, capture_output=True, text=True,
                         timeout=300)
    assert out.returncode == 0 and "OK" in out.stdout, out.stderr[-2000:]
