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


def test_summary_is_compact_and_complete():
    """`summary` - the LAST key of the line - must show [ms, roofline fraction, oracle parity] of the headline and of every
    secondary in well under 1 KB (a record that keeps only the tail of the line still shows every config), flag a
    secondary that failed, and say "differs" when a graph replay did not reproduce its eager step."""
    sec = {"c3_marginal": {"ms_per_step": 265.9031, "frac": 0.72118, "parity": {"residual_rel_err": 1.2204995e-07}},
           "c3_strong_rank8": {"joint": {"ms_per_step": 30.2199, "frac": 0.71955}, "marginal": {"ms_per_step": 32.0178, "frac": 0.72692},
                               "parity": {"residual_rel_err": 1.2048e-07}},
           "c3_rank8_ntfast": {"joint": {"ms_per_step": 31.9668, "frac": 0.70617}, "marginal": {"ms_per_step": 34.0884, "frac": 0.66932},
                               "parity": {"residual_rel_err": 1.2086e-07}},
           "c4_marginal_rank8": {"ms_per_step": 20.367, "frac": 0.712, "parity": {"residual_rel_err": 2.328e-07}},
           "c4_marginal_rank8_ntfast": {"ms_per_step": 20.553, "frac": 0.694, "parity": {"residual_rel_err": 1.7517e-07}},
           "c1": {"ms_per_step": 0.0927, "frac": 0.11, "parity": {"residual_rel_err": 0.0}},
           "c2": {"ms_per_step": 2.91, "frac": 0.67, "parity": {"residual_rel_err": 0.0}},
           "c4_shard": {"ms_per_step": 18.9, "frac": 0.706, "parity": {"residual_rel_err": 2.33e-07}},
           "c4_continuity": {"error": "RuntimeError: boom"},
           "c4_momentum": {"ms_per_step": 25.1, "frac": 0.688, "parity": {"residual_rel_err": 1.38e-07}},
           "c4_energy": {"ms_per_step": 28.7, "frac": 0.688, "parity": {"residual_rel_err": 1.81e-07}},
           "c4_gauss": {"ms_per_step": 12.8, "frac": 0.71, "parity": {"residual_rel_err": 0.0}},
           "c4_shard_ntfast": {"ms_per_step": 18.6, "frac": 0.703, "parity": {"residual_rel_err": 1.75e-07}},
           "c5_shard": {"ms_per_step": 2.23, "frac": 0.694, "parity": {"residual_rel_err": 5.7e-08}},
           "c5_whole": {"ms_per_step": 15.8, "frac": 0.722, "parity": {"residual_rel_err": 1.13e-07}},
           "c1_graph": {"ms_per_step_graph": 0.03706, "replay_equals_eager": True},
           "c2_graph": {"ms_per_step_graph": 2.4474, "replay_equals_eager": False},
           "c5_graph": {"error": "capture failed"}}
    out = {"ms_per_step": 242.78265, "roofline": {"frac": 0.7189119}, "parity": {"residual_rel_err": 1.22e-07}, "secondary": sec}
    s = bench.summary_of(out)
    text = json.dumps(s)
    assert len(text) < 1000, len(text)
    assert s["c3"] == [243.0, 0.719, 1.22e-07] and s["c3_marg"] == [266.0, 0.721, 1.22e-07]
    assert s["c3_r8_j"][:2] == [30.2, 0.72] and s["c3_r8_nt_m"][:2] == [34.1, 0.669] and s["c4_marg_nt"][0] == 20.6
    assert s["c4_cont"] == "error" and s["c5_g"] == "error" and s["c2_g"] == "differs" and s["c1_g"] == [0.0371]
    for key in ("c1", "c2", "c4_ind", "c4_mom", "c4_en", "c4_gauss", "c4_ind_nt", "c4_marg", "c5", "c5_whole"):
        assert len(s[key]) == 3, key
    assert bench.summary_of({"ms_per_step": 1.0, "roofline": {"frac": 0.5}}) == {"c3": [1.0, 0.5, None]}     # N > 1: no secondaries
    assert bench.med([3.0, 1.0, 47.0, 2.0, 2.5]) == 2.5 and bench.med([1.0, 3.0]) == 2.0


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
