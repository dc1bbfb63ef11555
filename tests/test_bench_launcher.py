"""``python bench.py --gpus N`` without torchrun: the parent process starts the
ranks itself (bench.launch_ranks) and never touches a GPU. CPU tests of the
rank / environment plumbing, of failure propagation, and -- on a box without a
GPU -- that two self-launched ranks rendezvous (gloo) and then stop exactly at
"no HIP device" (the product has no CPU path)."""

import json
import os
import subprocess
import sys
import textwrap
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_importing_bench_does_not_import_torch():
    code = "import sys; sys.path.insert(0, %r); import bench; assert 'torch' not in sys.modules" % ROOT
    subprocess.run([sys.executable, "-c", code], check=True)


def test_launch_ranks_exports_torchrun_environment(tmp_path):
    script = tmp_path / "child.py"
    script.write_text(textwrap.dedent("""
        import json, os, sys
        rec = {k: os.environ.get(k) for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE",
                                              "MASTER_ADDR", "MASTER_PORT", "HSA_ENABLE_IPC_MODE_LEGACY")}
        rec["argv"] = sys.argv[1:]
        open(os.path.join(os.path.dirname(__file__), f"rank{rec['RANK']}.json"), "w").write(json.dumps(rec))
        print("noise before the line")
        print(json.dumps({"rank": rec["RANK"]}))
    """))
    rc, out = bench.launch_ranks(3, ["--gpus", "3", "--steps", "1"], script=str(script))
    assert rc == 0
    assert json.loads(out.strip().splitlines()[-1]) == {"rank": "0"}  # only rank 0's stdout comes back
    recs = [json.loads((tmp_path / f"rank{r}.json").read_text()) for r in range(3)]
    for r, rec in enumerate(recs):
        assert rec["RANK"] == rec["LOCAL_RANK"] == str(r)
        assert rec["WORLD_SIZE"] == rec["LOCAL_WORLD_SIZE"] == "3"
        assert rec["MASTER_ADDR"] == "127.0.0.1"
        assert rec["HSA_ENABLE_IPC_MODE_LEGACY"] == os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        assert rec["argv"] == ["--gpus", "3", "--steps", "1"]
    assert len({rec["MASTER_PORT"] for rec in recs}) == 1 and int(recs[0]["MASTER_PORT"]) > 0


def test_launch_ranks_stops_the_others_when_one_fails(tmp_path):
    script = tmp_path / "child.py"
    script.write_text(textwrap.dedent("""
        import os, sys, time
        if os.environ["RANK"] == "1":
            sys.exit(3)
        time.sleep(120)
    """))
    t0 = time.monotonic()
    rc, _ = bench.launch_ranks(3, [], script=str(script))
    assert rc == 3
    assert time.monotonic() - t0 < 30


def test_launch_ranks_timeout(tmp_path):
    script = tmp_path / "child.py"
    script.write_text("import time; time.sleep(120)\n")
    rc, _ = bench.launch_ranks(2, [], script=str(script), timeout=1.0)
    assert rc == 124


def test_rank_environment_is_a_copy():
    base = {"A": "1"}
    env = bench.rank_environment(2, 4, 1234, base)
    assert base == {"A": "1"} and env["A"] == "1" and env["RANK"] == "2" and env["MASTER_PORT"] == "1234"


def test_strong_and_weak_arguments():
    a = bench.parse_args(["--gpus", "8", "--scaling", "strong", "--num-envs", "65536", "--recurrent", "--horizon", "256"])
    assert a.scaling == "strong" and a.num_envs == 65536 and a.recurrent and a.backend == "nccl"
    assert bench.parse_args([]).scaling == "weak" and bench.parse_args([]).gpus == 1


def test_self_launched_ranks_rendezvous_then_stop_at_no_hip_device():
    import torch

    if torch.cuda.is_available():
        pytest.skip("needs a box without a GPU (the GPU run is tests/test_sharded_gpu.py)")
    proc = subprocess.run(
        [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--num-envs", "64",
         "--steps", "1", "--warmup", "0"],
        capture_output=True, text=True, timeout=600,
        env={k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")},
    )
    assert proc.returncode != 0
    assert "rank 0/2 up, backend gloo" in proc.stderr and "rank 1/2 up, backend gloo" in proc.stderr
    assert "HipExtensionError" in proc.stderr and "needs a HIP device" in proc.stderr
    assert proc.stdout.strip() == ""  # no JSON line from a run that measured nothing
