"""`python bench.py --gpus N` as the driver types it: the parent starts the ranks under torch.distributed.run
itself (before touching torch.cuda / HIP), relays rank 0's single JSON line and the exit code.  CPU only: the
command that would be started (`--dry-run-launch`) and a gloo rehearsal of launcher -> ranks -> relay."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(*args, env=None):
    e = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, BENCH] + list(args), capture_output=True, text=True, env=e, timeout=300)


def test_dry_run_prints_the_launch_command():
    proc = _run("--gpus", "8", "--steps", "7", "--warmup", "2", "--dry-run-launch")
    assert proc.returncode == 0, proc.stderr
    cmd = json.loads(proc.stdout)["launch"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"]
    assert cmd[cmd.index("--nproc-per-node") + 1] == "8"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert int(cmd[cmd.index("--master-port") + 1]) > 1024
    tail = cmd[cmd.index(BENCH) + 1:]
    assert tail == ["--gpus", "8", "--steps", "7", "--warmup", "2"]           # the ranks get the same arguments


def test_single_gpu_run_needs_no_launcher():
    proc = _run("--dry-run-launch")
    assert proc.returncode == 0 and json.loads(proc.stdout)["launch"] is None


def test_under_a_launcher_nothing_is_started_again():
    """WORLD_SIZE set = torch.distributed.run already made the ranks: the script must go on as a rank (here it
    then stops for want of a GPU -- what matters is that it did not try to launch)."""
    proc = _run("--gpus", "2", "--steps", "1", "--warmup", "0", env={"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert "starting 2 ranks" not in proc.stderr
    assert proc.returncode != 0 and "needs an MI355X" in proc.stderr


def test_the_parent_never_imports_torch_before_launching():
    """Static: launch_ranks() and the code main() runs before it must not import torch (a HIP context in the
    parent would sit beside rank 0's on GPU 0)."""
    import ast
    src = open(BENCH).read()
    tree = ast.parse(src)
    for fn in (n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in ("launch_ranks", "launch_command", "free_port")):
        imported = [a.name for n in ast.walk(fn) if isinstance(n, (ast.Import, ast.ImportFrom))
                    for a in n.names] + [n.module for n in ast.walk(fn) if isinstance(n, ast.ImportFrom)]
        assert not [m for m in imported if m and (m.startswith("torch") or m.startswith("cobbletrace_amd"))], (fn.name, imported)
    top = [a.name for n in tree.body if isinstance(n, ast.Import) for a in n.names]
    assert "torch" not in top                                             # nothing at module level either
    main = src[src.index("def main():"):]
    assert main.index("launch_ranks(args") < main.index("import torch")


def test_gloo_rehearsal_of_launch_and_relay():
    proc = _run("--gpus", "3", "--rehearse-launch")
    assert proc.returncode == 0, proc.stderr
    lines = [l for l in proc.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, proc.stdout                       # exactly ONE line on stdout: the record
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 3 and rec["ranks"] == [[0, 100.0], [1, 101.0], [2, 102.0]]
    assert "not the record" in proc.stderr                    # the other ranks' stdout noise went to stderr


def test_a_failing_rank_fails_the_parent():
    """No GPU here: the ranks stop with an error, and the parent must hand that exit code on, not a record."""
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("a GPU is present")
    proc = _run("--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu-baseline")
    assert proc.returncode != 0 and proc.stdout.strip() == ""
