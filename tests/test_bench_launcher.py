"""bench.py --gpus N: the job size is what the line reports.  Host logic only (no GPU): N > 1 without a torchrun environment
spawns N ranks, a WORLD_SIZE that disagrees with --gpus is an error, --gpus 1 / a matching torchrun environment run in place."""
import importlib.util
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_launch_plan():
    b = _bench()
    assert b.launch_plan(1, {}, ["--gpus", "1"]) == ("run", None, 1)
    assert b.launch_plan(None, {}, []) == ("run", None, 1)
    assert b.launch_plan(None, {"WORLD_SIZE": ""}, []) == ("run", None, 1)          # an empty value is "not set"
    assert b.launch_plan(8, {"WORLD_SIZE": "8", "RANK": "3"}, ["--gpus", "8"]) == ("run", None, 8)
    what, msg, _n = b.launch_plan(8, {"WORLD_SIZE": "1"}, ["--gpus", "8"])
    assert what == "error" and "WORLD_SIZE=1" in msg
    what, msg, _n = b.launch_plan(1, {"WORLD_SIZE": "4"}, ["--gpus", "1"])          # an EXPLICIT --gpus that disagrees
    assert what == "error"
    # `torchrun --nproc-per-node 4 bench.py` without --gpus: the job size is WORLD_SIZE (an external scaling driver may launch that way)
    assert b.launch_plan(None, {"WORLD_SIZE": "4", "RANK": "1"}, ["--steps", "3"]) == ("run", None, 4)
    what, cmd, n = b.launch_plan(4, {}, ["--gpus", "4", "--steps", "3", "--warmup", "1"])
    assert what == "spawn" and n == 4
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nproc-per-node=4" in cmd and "--nnodes=1" in cmd
    # torchrun picks the rendezvous port itself (no pick-then-bind race) and the workers meet on the loopback address
    assert "--standalone" in cmd and cmd[cmd.index("--local-addr") + 1] == "127.0.0.1" and "--master-port" not in cmd
    i = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[i + 1:] == ["--gpus", "4", "--steps", "3", "--warmup", "1"]


def test_standalone_child_runs_two_ranks_on_the_loopback_address(tmp_path):
    """The spawn form itself, with a stand-in for bench.py: two gloo ranks rendezvous through `--standalone --local-addr 127.0.0.1`."""
    b = _bench()
    child = tmp_path / "rk.py"
    child.write_text("import os, sys, torch.distributed as dist\ndist.init_process_group('gloo')\n"
                     "open(os.path.join(sys.argv[1], f'rank{dist.get_rank()}'), 'w').write(f\"{dist.get_world_size()} {os.environ['MASTER_ADDR']}\")\n"
                     "dist.barrier()\ndist.destroy_process_group()\n")
    _what, cmd, _n = b.launch_plan(2, {}, [str(tmp_path)])
    cmd[cmd.index(os.path.join(ROOT, "bench.py"))] = str(child)
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=dict(os.environ, GLOO_SOCKET_IFNAME="lo"))
    assert r.returncode == 0, r.stderr[-2000:]
    assert (tmp_path / "rank0").read_text() == "2 127.0.0.1" and (tmp_path / "rank1").read_text() == "2 127.0.0.1"


def test_world_size_mismatch_exits_nonzero_before_touching_the_gpu():
    env = dict(os.environ, WORLD_SIZE="2", RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=2" in (r.stderr + r.stdout)


def test_spawn_relays_output_and_keeps_the_json_line_last(tmp_path):
    b = _bench()
    child = tmp_path / "child.py"
    child.write_text("import sys\nprint('banner')\nprint('{\"metric\": \"m\", \"value\": 1}')\nprint('trailing noise')\nsys.exit(3)\n")
    r = subprocess.run([sys.executable, "-c",
                        f"import sys; sys.path.insert(0, {ROOT!r}); import importlib.util as u; s = u.spec_from_file_location('b', {os.path.join(ROOT, 'bench.py')!r}); "
                        f"m = u.module_from_spec(s); s.loader.exec_module(m); sys.exit(m.spawn_ranks([sys.executable, {str(child)!r}]))"],
                       capture_output=True, text=True, timeout=300)
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert r.returncode == 3
    assert lines[0] == "banner" and lines[-1] == '{"metric": "m", "value": 1}'


def test_bench_gpus_4_starts_four_ranks_that_refuse_to_run_without_a_gpu():
    """The real bench.py as the launcher of a 4-rank job (gloo, as the one-GPU box shares its device): here, without a GPU, every
    rank that gets to run must stop with the "no CPU fallback" message and the launcher must hand the failure on - the product path never
    degrades to a CPU run.  (The same command on the GPU box runs four ranks on the shared device: tests/test_gpu_dp.py.)"""
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("this is the no-GPU leg; tests/test_gpu_dp.py runs the command for real")
    env = dict(os.environ, SEG_DIST_BACKEND="gloo", GLOO_SOCKET_IFNAME="lo")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--workload", "tiny", "--steps", "1", "--warmup", "0",
                        "--no-extras", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode != 0
    # (torchrun tears the other workers down as soon as the first one has failed: between one and four of them get to say it)
    assert 1 <= (r.stderr + r.stdout).count("bench.py needs an MI355X (no CPU fallback)") <= 4
    assert "--nproc-per-node=4" in " ".join(_bench().launch_plan(4, {}, ["--gpus", "4"])[1])
    assert '"metric"' not in r.stdout
