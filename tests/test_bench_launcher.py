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
    assert b.launch_plan(1, {}, ["--gpus", "1"]) == ("run", None)
    assert b.launch_plan(8, {"WORLD_SIZE": "8", "RANK": "3"}, ["--gpus", "8"]) == ("run", None)
    what, msg = b.launch_plan(8, {"WORLD_SIZE": "1"}, ["--gpus", "8"])
    assert what == "error" and "WORLD_SIZE=1" in msg
    what, msg = b.launch_plan(1, {"WORLD_SIZE": "4"}, [])
    assert what == "error"
    what, cmd = b.launch_plan(4, {}, ["--gpus", "4", "--steps", "3", "--warmup", "1"], lambda: 12345)
    assert what == "spawn"
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nproc-per-node=4" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "12345"
    i = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[i + 1:] == ["--gpus", "4", "--steps", "3", "--warmup", "1"]


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
