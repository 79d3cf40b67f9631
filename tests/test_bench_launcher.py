"""bench.py --gpus N: the launcher logic (decided before any GPU call) and the scaling flag.  CPU only;
the spawned ranks run `--plumbing-check` (process group + one all-reduce over gloo, no compute)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_launch_plan():
    assert bench.launch_plan(1, {}, []) == ("run", 1)
    kind, cmd = bench.launch_plan(8, {}, ["--gpus", "8", "--steps", "2"], script="bench.py")
    assert kind == "spawn"
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--standalone"] and "--nproc-per-node" in cmd
    assert cmd[cmd.index("--nproc-per-node") + 1] == "8" and cmd[-5:] == ["bench.py", "--gpus", "8", "--steps", "2"]
    assert "127.0.0.1" in cmd
    assert bench.launch_plan(8, {"WORLD_SIZE": "8"}, []) == ("run", 8)
    assert bench.launch_plan(1, {"WORLD_SIZE": "1"}, []) == ("run", 1)
    for gpus, ws in ((8, "1"), (1, "2"), (4, "8")):               # a wrong-size run must never produce a number
        kind, msg = bench.launch_plan(gpus, {"WORLD_SIZE": ws}, [])
        assert kind == "error" and ws in msg and str(gpus) in msg


def test_split_slabs():
    assert bench.split_slabs(64, 13) == [13, 13, 13, 13, 12]
    assert bench.split_slabs(64, 8) == [8] * 8
    assert bench.split_slabs(64, 16) == [16] * 4
    assert bench.split_slabs(64, 64) == [64] and bench.split_slabs(10, 64) == [10]
    for nt in (1, 7, 62, 64, 200):
        for s in (1, 5, 12, 13, 100):
            parts = bench.split_slabs(nt, s)
            assert sum(parts) == nt and max(parts) <= s and max(parts) - min(parts) <= 1


def _run(argv, env_extra=None, timeout=240):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], env=env, capture_output=True,
                          text=True, timeout=timeout)


@pytest.mark.timeout(300)
def test_bare_gpus_2_spawns_two_ranks():
    """`python bench.py --gpus 2` with WORLD_SIZE unset starts two ranks itself and reports n_gpus 2."""
    r = _run(["--gpus", "2", "--plumbing-check", "--scaling", "strong", "--batch", "256"])
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 2 and out["rccl_ranks"] == 2
    assert out["scaling"] == "strong" and out["batch_per_rank"] == 128


def test_world_size_mismatch_exits_2():
    r = _run(["--gpus", "8", "--plumbing-check"], {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode == 2 and "WORLD_SIZE=1" in r.stderr
    r = _run(["--gpus", "2", "--plumbing-check", "--scaling", "strong", "--batch", "255"],
             {"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode == 2 and "divisible" in r.stderr


def test_c3_interior_row_slabs_and_c4_equations():
    """The C3 job streams the reference's interior rows 1 .. Nx-2 (Marginal/NS_Residuals_CP.py:240): 510 rows in runs of 128
    are 128, 128, 127, 127 - the same bytes resident as for 4 x 128; and each of C4's five equations
    (Marginal/MHD_Residuals_CP.py:225-278) has its bytes per cell = 4 (fields read + 1) and a kernel the library instantiates."""
    assert bench.split_slabs(510, 128) == [128, 128, 127, 127]
    assert bench.split_slabs(510, 510) == [510] and bench.split_slabs(510, 255) == [255, 255]
    other = 64 * 512
    assert bench.resident_bytes(4096, 510, 128, other) == ((4096 + 3) * 3 * 130 + 4096 * 128) * other * 4 + 4096 * 256
    fields = {"continuity": 3, "momentum": 6, "energy": 6, "induction": 4, "gauss": 2}
    src = open(os.path.join(ROOT, "cp_pre_amd", "csrc", "star_march.hip")).read()
    for eq, f in fields.items():
        cfg = bench.mhd_config(eq)
        assert cfg["bpc"] == 4 * (f + 1) and cfg["equation"] == eq and cfg["shape"] == (1024, 64, 256, 256)
        functor = cfg["kernel"].split("<")[1].split("<")[0].split(",")[0]
        assert f"struct {functor} " in src or f"struct {functor}\n" in src, functor
