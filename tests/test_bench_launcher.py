"""bench.py's rank launcher (`python bench.py --gpus N` starts its own N ranks) on the CPU: the parts that cannot be exercised with
more than one rank on a single-GPU test box - command line, environment, relay of rank 0's JSON line as the ONLY stdout line,
exit-code propagation, the refusal when the node has fewer GPUs, the sanity check that the line really describes an N-rank run."""
import importlib.util
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture()
def bench(monkeypatch):
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


class _Proc:
    def __init__(self, stdout, rc):
        self._out, self.returncode, self.pid = stdout, rc, 12345

    def communicate(self, timeout=None):
        return self._out, None


def _patch(monkeypatch, bench, n_dev, stdout, rc, seen):
    import subprocess
    monkeypatch.setattr(bench.torch.cuda, "device_count", lambda: n_dev)

    def popen(cmd, **kw):
        seen["cmd"], seen["env"], seen["kw"] = cmd, kw.get("env"), kw
        return _Proc(stdout, rc)
    monkeypatch.setattr(subprocess, "Popen", popen)


def _line(n, ranks=None):
    return json.dumps({"metric": "tri-modal images/sec fwd+bwd @ B=128 ViT-B", "value": 1.0, "n_gpus": n, "rccl_ranks": n if ranks is None else ranks})


def test_launcher_command_environment_and_relay(bench, monkeypatch, capsys):
    seen = {}
    out = "NCCL version banner\n[rank3] something\n" + _line(4) + "\ntrailing noise\n"
    _patch(monkeypatch, bench, 8, out, 0, seen)
    rc = bench.launch_ranks(4, ["--gpus", "4", "--steps", "5", "--spawn", "--warmup", "2"])
    assert rc == 0
    cmd = seen["cmd"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and int(cmd[cmd.index("--master-port") + 1]) > 1024
    tail = cmd[cmd.index(os.path.join(ROOT, "bench.py")) + 1:]
    assert tail == ["--gpus", "4", "--steps", "5", "--warmup", "2"]          # same argv, without --spawn
    env = seen["env"]
    assert env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" and "EDITOR_FORCE_DDP" not in env
    assert seen["kw"].get("start_new_session") is True                        # its own process group: a timeout kills only that
    cap = capsys.readouterr()
    assert cap.out.strip().splitlines() == [_line(4)]                         # the JSON line is the ONLY stdout line
    assert "NCCL version banner" in cap.err and "trailing noise" in cap.err   # everything else goes to stderr


def test_launcher_one_rank_forces_a_real_process_group(bench, monkeypatch, capsys):
    seen = {}
    _patch(monkeypatch, bench, 1, _line(1) + "\n", 0, seen)
    assert bench.launch_ranks(1, ["--gpus", "1", "--spawn"]) == 0
    assert seen["env"]["EDITOR_FORCE_DDP"] == "1" and "--nproc-per-node=1" in seen["cmd"]


def test_launcher_refuses_more_ranks_than_gpus(bench, monkeypatch, capsys):
    seen = {}
    _patch(monkeypatch, bench, 2, "", 0, seen)
    assert bench.launch_ranks(8, ["--gpus", "8"]) != 0
    assert "cmd" not in seen                                                  # nothing was started
    cap = capsys.readouterr()
    assert cap.out == "" and "8" in cap.err and "2 GPU" in cap.err


@pytest.mark.parametrize("stdout,rc", [("", 1), (_line(4), 7), ("no json at all\n", 0), (_line(4, ranks=1), 0), (_line(2), 0)])
def test_launcher_failures_are_loud(bench, monkeypatch, capsys, stdout, rc):
    """a crashed rank (rc != 0), no JSON line, or a line that does not describe a 4-rank RCCL run: non-zero exit, empty stdout"""
    seen = {}
    _patch(monkeypatch, bench, 8, stdout, rc, seen)
    got = bench.launch_ranks(4, ["--gpus", "4"])
    assert got != 0 and (rc == 0 or got == rc)
    assert capsys.readouterr().out == ""


def test_rank_process_checks_its_world_size(bench, monkeypatch):
    monkeypatch.setenv("WORLD_SIZE", "2")
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4"])
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert "WORLD_SIZE=2" in str(e.value)


def test_launcher_failure_shows_every_ranks_stderr_tail(bench, monkeypatch, capsys, tmp_path):
    """a failed N-rank run prints each rank's last stderr lines (torch.distributed.run --log-dir / --tee 2),
    and a rank-prefixed JSON line (teed stdout) is still recognised."""
    import tempfile
    seen = {}
    _patch(monkeypatch, bench, 8, "", 1, seen)
    for r in range(2):
        d = tmp_path / "run_x" / "attempt_0" / str(r)
        d.mkdir(parents=True)
        (d / "stderr.log").write_text("".join("rank %d line %d\n" % (r, i) for i in range(40)))
    monkeypatch.setattr(tempfile, "mkdtemp", lambda prefix="": str(tmp_path))
    assert bench.launch_ranks(2, ["--gpus", "2"]) != 0
    cmd = seen["cmd"]
    assert cmd[cmd.index("--log-dir") + 1] == str(tmp_path) and cmd[cmd.index("--tee") + 1] == "2"
    err = capsys.readouterr().err
    assert "rank 0 line 39" in err and "rank 1 line 39" in err and "rank 0 line 3\n" not in err
    seen = {}
    _patch(monkeypatch, bench, 8, "[default0]:" + _line(2) + "\n", 0, seen)
    assert bench.launch_ranks(2, ["--gpus", "2"]) == 0
    assert capsys.readouterr().out.strip() == _line(2)


def test_committed_profile_is_quoted_only_for_the_tree_it_was_taken_on(bench, monkeypatch, tmp_path):
    """bench.py's `in_situ_*` figures come from the newest profiles/rNN_bench_kernel_stats_serial.csv ONLY when the
    source hash stamped beside it (tools/prof.sh -> .hash) equals the running tree's; otherwise nothing is quoted and the line says why."""
    prof = tmp_path / "profiles"
    prof.mkdir()
    (prof / "r07_bench_kernel_stats_serial.csv").write_text(
        '"Name","Calls","TotalDurationNs","AverageNs","Percentage","MinNs","MaxNs","StdDev"\n'
        '"void (anonymous namespace)::layernorm_fwd_kernel<x>(a)",10,1000,35500.0,1,1,1,1\n')
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    monkeypatch.setattr(bench, "source_hash", lambda: "aaaa")
    src, situ = bench._in_situ_us()                       # no stamp at all
    assert situ == {} and "NOT quoted" in src
    (prof / "r07_bench_kernel_stats_serial.hash").write_text("bbbb\n")
    src, situ = bench._in_situ_us()                       # stamp of another tree
    assert situ == {} and "NOT quoted" in src
    (prof / "r07_bench_kernel_stats_serial.hash").write_text("aaaa\n")
    src, situ = bench._in_situ_us()
    assert src.endswith("r07_bench_kernel_stats_serial.csv") and abs(situ["layernorm_fwd_kernel<x>(a)"] - 35.5) < 1e-9


def test_source_hash_moves_with_a_kernel_source(bench, monkeypatch, tmp_path):
    import shutil
    root = tmp_path / "r"
    for d in ("editor_amd/csrc", "include", "editor_amd/modeling"):
        (root / d).mkdir(parents=True)
    (root / "editor_amd/csrc/a.hip").write_text("kernel 1")
    (root / "include/x.h").write_text("h")
    monkeypatch.setattr(bench, "ROOT", str(root))
    h0 = bench.source_hash()
    assert h0 == bench.source_hash() and len(h0) == 16
    (root / "editor_amd/csrc/a.hip").write_text("kernel 2")
    assert bench.source_hash() != h0
