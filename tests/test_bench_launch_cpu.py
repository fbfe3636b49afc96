"""bench.py --gpus N: the launcher logic that must run before anything touches the GPU (VERDICT r1 #1,
ADVICE r1): a plain `python bench.py --gpus N` starts N ranks as a child torchrun; a WORLD_SIZE that
disagrees with --gpus is refused instead of silently benchmarking a different rank count."""
import os
import subprocess
import sys
import types

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load_bench():
    import importlib.util

    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_world_size_mismatch_is_refused():
    env = dict(os.environ, WORLD_SIZE="3", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env,
                       capture_output=True, text=True, timeout=120)
    assert p.returncode == 2
    assert "WORLD_SIZE=3" in p.stderr and p.stdout.strip() == ""


def test_self_launch_builds_a_child_torchrun(monkeypatch):
    bench = _load_bench()
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 7

    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.delenv("CORSAIR_DIST_BACKEND", raising=False)
    monkeypatch.setattr(bench.subprocess, "call", fake_call)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3", "--workload", "table"])
    with pytest.raises(SystemExit) as e:
        bench.maybe_self_launch(types.SimpleNamespace(gpus=4))
    assert e.value.code == 7                      # the child's return code is ours
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nproc-per-node" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "4"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-6:] == ["--gpus", "4", "--steps", "3", "--workload", "table"]
    # no GPU in the build container: fewer devices than ranks -> the ranks share devices over gloo
    assert seen["env"]["CORSAIR_DIST_BACKEND"] == "gloo"


def test_single_gpu_and_torchrun_ranks_do_not_relaunch(monkeypatch):
    bench = _load_bench()
    monkeypatch.setattr(bench.subprocess, "call", lambda *a, **k: pytest.fail("must not launch"))
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    bench.maybe_self_launch(types.SimpleNamespace(gpus=1))
    monkeypatch.setenv("WORLD_SIZE", "8")
    bench.maybe_self_launch(types.SimpleNamespace(gpus=8))


def test_table_label_histogram():
    bench = _load_bench()
    lab = bench.sym_labels("table", 830)
    vals, cnt = np.unique(lab, return_counts=True)
    assert dict(zip(vals.tolist(), cnt.tolist())) == {1: 233, 2: 422, 3: 7, 4: 128, 12: 40}
    chair = bench.sym_labels("chair", 652)
    assert (chair == 1).sum() == 650 and (chair == 4).sum() == 2
    assert len(bench.sym_labels("table", 64)) == 64


def test_batches_in_flight_pass_is_skipped_when_a_step_holds_a_collective():
    """The crash fixed in 86a7de4 (`--workload stress --gpus 2`): the extra three-batches-in-flight pass drives the steps
    from three host threads; with an all-gather inside the step the ranks' collectives mis-pair.  The decision is one
    function: never with a collective in the step on several ranks, never for strong-scaling steps."""
    bench = _load_bench()
    plain = types.SimpleNamespace()
    coll = types.SimpleNamespace(collective_in_step=True)
    strong = types.SimpleNamespace(collective_in_step=True, scaling="strong")
    assert bench.overlap_probe_allowed(1, 8, False, 1, plain)
    assert bench.overlap_probe_allowed(1, 8, False, 8, plain)          # chair / table: no collective in the step
    assert bench.overlap_probe_allowed(1, 8, False, 1, coll)           # one rank: nothing to mis-pair
    assert not bench.overlap_probe_allowed(1, 8, False, 2, coll)       # stress on 2 ranks
    assert not bench.overlap_probe_allowed(1, 8, False, 1, strong) and not bench.overlap_probe_allowed(1, 8, False, 4, strong)
    assert not bench.overlap_probe_allowed(3, 8, False, 1, plain) and not bench.overlap_probe_allowed(1, 1, False, 1, plain)
    assert not bench.overlap_probe_allowed(1, 8, True, 1, plain)
    assert bench.StressWorkload.collective_in_step and bench.StrongEvalWorkload.collective_in_step
    assert not getattr(bench.RegistrationWorkload, "collective_in_step", False)
