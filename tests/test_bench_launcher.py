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


def test_launch_timeout_is_finite_by_default():
    a = bench.parse_args(["--gpus", "8"])
    assert a.launch_timeout is None                       # -> default_launch_timeout(steps, warmup) in main()
    limit = bench.default_launch_timeout(a.steps, a.warmup)
    assert 600 < limit < 3600 and bench.default_launch_timeout(20, 5) > limit
    assert bench.parse_args(["--launch-timeout", "30"]).launch_timeout == 30.0


def test_failed_rank_stderr_tail_is_forwarded(tmp_path, capfd):
    script = tmp_path / "child.py"
    script.write_text(textwrap.dedent("""
        import os, sys, time
        print(f"hello from rank {os.environ['RANK']}", file=sys.stderr, flush=True)
        if os.environ["RANK"] == "1":
            print("RuntimeError: boom in rank one", file=sys.stderr, flush=True)
            sys.exit(7)
        time.sleep(120)
    """))
    rc, _ = bench.launch_ranks(2, [], script=str(script))
    err = capfd.readouterr().err
    assert rc == 7
    assert "[rank 1] RuntimeError: boom in rank one" in err            # passed through live, tagged
    assert "last lines of rank 1's stderr" in err and "last lines of rank 0's stderr" in err
    assert err.count("boom in rank one") >= 2                           # ... and again in the tail


def _alive(pid: int) -> bool:
    try:
        os.kill(pid, 0)
    except ProcessLookupError:
        return False
    try:  # a zombie still answers kill(0)
        with open(f"/proc/{pid}/stat") as f:
            return f.read().split(") ")[1][0] != "Z"
    except OSError:
        return False


@pytest.mark.parametrize("sig", ["SIGTERM", "SIGKILL"])
def test_no_rank_survives_a_killed_parent(tmp_path, sig):
    """SIGTERM: the launcher's handler stops the ranks' process groups. SIGKILL (no handler can run): every rank asked
    the kernel for SIGTERM on parent death (bench.die_with_parent, first thing in a launched rank; it also leaves if
    the launcher is already gone by then). Either way nothing is left holding a GPU."""
    import signal

    child = tmp_path / "child.py"
    child.write_text(textwrap.dedent(f"""
        import os, subprocess, sys, time
        sys.path.insert(0, {ROOT!r})
        import bench
        bench.die_with_parent(int(os.environ["RL8_BENCH_PARENT"]))  # what bench.main() does first in a launched rank
        helper = subprocess.Popen([sys.executable, "-c", "import time; time.sleep(300)"])  # a rank's own helper process
        open(os.path.join(os.path.dirname(__file__), f"pid{{os.environ['RANK']}}"), "w").write(f"{{os.getpid()}} {{helper.pid}}")
        time.sleep(300)
    """))
    parent_code = f"import sys; sys.path.insert(0, {ROOT!r}); import bench; bench.launch_ranks(3, [], script={str(child)!r})"
    parent = subprocess.Popen([sys.executable, "-c", parent_code], stderr=subprocess.DEVNULL)
    try:
        deadline = time.monotonic() + 60
        while time.monotonic() < deadline and not all((tmp_path / f"pid{r}").exists() and (tmp_path / f"pid{r}").read_text().count(" ") for r in range(3)):
            time.sleep(0.05)
        pids = [[int(v) for v in (tmp_path / f"pid{r}").read_text().split()] for r in range(3)]
        assert all(_alive(p[0]) for p in pids)
        parent.send_signal(getattr(signal, sig))
        parent.wait(timeout=60)
        if sig == "SIGTERM":
            assert parent.returncode == 128 + signal.SIGTERM
        deadline = time.monotonic() + 30
        ranks = [p[0] for p in pids]
        while time.monotonic() < deadline and any(_alive(p) for p in ranks):
            time.sleep(0.1)
        assert not any(_alive(p) for p in ranks), "a rank outlived its launcher"
        if sig == "SIGTERM":  # the handler signals the whole process GROUP of each rank: helpers go too
            helpers = [p[1] for p in pids]
            while time.monotonic() < deadline and any(_alive(p) for p in helpers):
                time.sleep(0.1)
            assert not any(_alive(p) for p in helpers)
    finally:
        if parent.poll() is None:
            parent.kill()
        for r in range(3):
            f = tmp_path / f"pid{r}"
            if f.exists():
                for p in f.read_text().split():
                    try:
                        os.kill(int(p), signal.SIGKILL)
                    except (ProcessLookupError, ValueError):
                        pass


def test_rccl_run_without_enough_devices_fails_before_any_rendezvous():
    import torch

    if torch.cuda.device_count() >= 2:
        pytest.skip("needs a box with fewer than two HIP devices")
    t0 = time.monotonic()
    proc = subprocess.run(
        [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--num-envs", "64"],
        capture_output=True, text=True, timeout=600,
        env={k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")},
    )
    assert proc.returncode != 0 and proc.stdout.strip() == ""
    assert "needs 2 HIP devices" in proc.stderr and "up, backend" not in proc.stderr   # no process group was formed
    assert "last lines of rank" in proc.stderr
    assert time.monotonic() - t0 < 300
