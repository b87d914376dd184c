"""CPU suite: `python bench.py --gpus N` started plainly (no WORLD_SIZE in the environment) must start the N
ranks itself and report N -- never quietly bench one GPU.  Driven here with the stub step (RT_BENCH_STUB=1:
gloo, no GPU; every step packs, gathers and assembles a frame with the host mirror of the shard layout)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env(**extra):
    env = {k: v for k, v in os.environ.items()
           if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID")}
    env.update(RT_BENCH_STUB="1", **extra)
    return env


def test_bench_starts_its_own_ranks_for_gpus_2():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--film", "200", "120"], env=_env(), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines                      # ONE JSON line, whatever the ranks printed
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 2 and line["warmup"] == 1
    assert line["data"] == "stub" and line["scaling"] == "strong" and line["value"] > 0
    # both ranks' tiles were counted: 7 x 4 tiles of 32 x 32 pixels cover the 200 x 120 film
    assert line["value"] * line["ms_per_step"] * 1e3 * 2 / 2 == __import__("pytest").approx(28 * 1024, rel=1e-2)


def test_bench_refuses_a_rank_count_that_differs_from_gpus():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "1", "--warmup", "0",
                        "--film", "64", "64"], env=_env(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 2 and r.stdout.strip() == ""
    assert "refusing" in r.stderr


def test_bench_reports_a_failing_rank():
    # a rank that dies must make the launcher exit non-zero without printing a line
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                        "--film", "64", "64"], env=_env(RT_BENCH_STUB_FAIL_RANK="1"), capture_output=True, text=True, timeout=600)
    assert r.returncode != 0 and r.stdout.strip() == ""


def test_bench_does_not_start_ranks_from_under_a_profiler():
    # rocprofv3's tool library has initialised the GPU before main() runs: no child processes from there
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=_env(ROCPROF_OUTPUT_PATH="x"),
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 2 and r.stdout.strip() == "" and "rocprofv3" in r.stderr
