"""bench.py prints ONE JSON line with the driver's contract fields plus `roofline`, `cpu_baseline` and the
parity sample (small workload here; the full-size line is the driver's) — and `--gpus 2` from a bare
invocation really runs two ranks (both on GPU 0 here, host-staged exchange) and reports both N > 1 modes."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*flags, timeout=900):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *flags], capture_output=True, text=True,
                         timeout=timeout, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, "exactly one line on stdout"
    return json.loads(lines[0])


def _contract(j, n_gpus, steps):
    for k, typ in (("metric", str), ("value", float), ("unit", str), ("n_gpus", int), ("steps", int), ("warmup", int),
                   ("ms_per_step", float), ("higher_is_better", bool), ("scaling", str), ("dtype", str), ("data", str),
                   ("config", dict), ("roofline", dict)):
        assert isinstance(j[k], typ), (k, j[k])
    assert j["vs_baseline"] is None and j["n_gpus"] == n_gpus and j["steps"] == steps and j["warmup"] == 1
    assert j["higher_is_better"] is True and j["scaling"] == "weak" and j["data"] == "synthetic" and j["value"] > 0
    assert "workload" in j["config"] and "model" not in j["config"]
    r = j["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and r["achieved"] > 0 and "traffic" in r
    if "shard_step_kernel" in r["kernel"]:          # --mode sharded: the frontier-step loop, no single dominant launch
        assert r["avg_launch_ms"] > 0
        return
    assert r["kernel"] in ("trav_kernel", "trav4_kernel") and 1 <= r["launches"] <= steps and r["avg_launch_ms"] > 0   # (--chain: several steps per launch)
    assert r["launch_ms_p10"] <= r["launch_ms_median"] <= r["launch_ms_p90"]


def test_bench_json_contract(gpu):
    j = _run("--gpus", "1", "--steps", "2", "--warmup", "1", "--rows", "300000", "--nq", "1024", "--n-to-score", "3000",
             "--cpu-seconds", "1")
    _contract(j, 1, 2)
    c = j["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and c["unit"] == j["unit"] and c["sample"]
    assert c["one_thread_value"] > 0 and c["cpu_model"]
    done, total = j["parity_sample"].split("/")
    assert done == total and int(total) > 0                      # the oracle's sample equals the GPU's counters
    assert j["config"]["corpus_mode"] == 2 and j["config"]["graph_recall_at_10_ef128"] > 0.5
    assert j["config"]["expansion_add"] == 400 and j["config"]["graph_recall_at_10_ef400"] >= j["config"]["graph_recall_at_10_ef128"]
    r = j["roofline"]
    # the two timed steps went into ONE launch (--chain): rows take the traversals of both from one counter, the scored lists live in
    # a ring that still holds the last step's (the parity sample above read them)
    assert r["launches"] == 1 and 0 < r["kernel_busy_ms"] <= r["avg_launch_ms"] * r["launches"] * 1.001
    ts = j["config"]["traversal_state"]
    assert ts["objects"] == 1 and ts["steps_per_launch"] == 2
    # ... and the two-object pipeline of overlapped launches (--chain 0) prints the same kind of line
    j0 = _run("--gpus", "1", "--steps", "3", "--warmup", "1", "--rows", "300000", "--nq", "1024", "--n-to-score", "3000",
              "--cpu-seconds", "1", "--chain", "0", "--no-config-legs", "--no-kernel-legs", "--secondary-expansion-add", "0")
    _contract(j0, 1, 3)
    assert j0["config"]["traversal_state"]["objects"] == 2 and j0["roofline"]["launches"] == 3
    assert j0["roofline"]["kernel_busy_ms"] <= j0["roofline"]["avg_launch_ms"] * 3 * 1.001
    done, total = j0["parity_sample"].split("/")
    assert done == total and int(total) > 0
    # the other kernels of the path, the rounds-1-3 graph and BASELINE configs[1] / [4] ride in the same line
    k = j["kernels"]
    assert k["scan_8q"]["GB/s"] > 0 and k["gather"]["GB/s"] > 0 and k["topk_8q_k10"]["self_is_nearest"] is True
    sg = j["secondary_graph"]
    assert sg["expansion_add"] == 64 and sg["value"] > 0 and 0 < sg["roofline_frac"] < 1
    c = j["configs"]
    assert set(c) == {"c1_1M_1024bit_m8", "c4_2M_2048bit_m32_ef400", "notebook_shape_2M_1024bit_m16_ef400"}
    for leg in c.values():
        assert leg.get("skipped") or (leg["value"] > 0 and 0 < leg["roofline_frac"] < 1)
    assert c["c4_2M_2048bit_m32_ef400"].get("skipped") or (c["c4_2M_2048bit_m32_ef400"]["ndim"] == 2048 and c["c4_2M_2048bit_m32_ef400"]["connectivity"] == 32)


def test_bench_two_ranks_from_a_bare_invocation(gpu):
    """no launcher: bench.py spawns its ranks itself; two ranks share GPU 0, so the exchange is host-staged.
    --mode sharded is the shard-native setup (every rank creates only its rows; the graph is built once on rank 0 and
    handed over); the default mode runs the replicas leg and the sharded leg after it."""
    j = _run("--gpus", "2", "--steps", "1", "--warmup", "1", "--rows", "200000", "--nq", "512", "--n-to-score", "1500",
             "--sharded-nq", "64", "--exchange", "host", "--single-device", "--mode", "sharded")
    _contract(j, 2, 1)
    sh = j["sharded"]
    assert j["value"] == sh["value"] > 0 and "replicas" not in j
    done, total = sh["parity_vs_single_gpu"].split("/")
    assert done == total and int(total) >= 2 * 64                    # counters of every sampled traversal + full scored lists
    assert sh["frontier_steps_per_step"] > 10 and sh["exchanged_bytes_per_rank_per_step"] > 0
    assert "row-sharded" in j["config"]["parallelism"] and "host-staged" in sh["exchange"]
    assert "shard-native" in sh["setup"] and len(sh["index_bytes_per_rank"]) == 2 and sh["rccl"] is None
    # a rank's index = its 100000 rows + the whole adjacency: less than the 200000 rows alone would take twice over
    assert max(sh["index_bytes_per_rank"]) < 100000 * 128 + 200000 * (16 * 4 + 5) * 1.6 + (1 << 20)
    # closed-form graph: no rank ever holds the corpus, not even to build
    j1 = _run("--gpus", "2", "--steps", "1", "--warmup", "1", "--rows", "200000", "--n-to-score", "1500", "--sharded-nq", "64",
              "--exchange", "host", "--single-device", "--mode", "sharded", "--graph", "synthetic", "--corpus-mode", "1",
              "--sharded-reference", "none")
    assert j1["sharded"]["parity_vs_single_gpu"] is None and "closed form" in j1["sharded"]["setup"] and j1["value"] > 0
    j2 = _run("--gpus", "2", "--steps", "1", "--warmup", "1", "--rows", "200000", "--nq", "512", "--n-to-score", "1500",
              "--sharded-nq", "64", "--exchange", "host", "--single-device")            # default: value from the replicas leg
    assert j2["value"] == j2["replicas"]["value"] and "replicas" in j2["config"]["parallelism"]
    done, total = j2["sharded"]["parity_vs_single_gpu"].split("/")
    assert done == total and int(total) >= 2 * 64                                         # the sharded leg still ran
    # ... and so did the peer-mapped leg: two rank processes, each with half of the rows, the other half imported through a dmabuf
    # descriptor that crossed a Unix socket; every sampled traversal equals the one over the rank's whole-corpus index
    pm = j2["peer_mapped"]
    assert pm.get("error") is None, pm
    done, total = pm["parity_vs_whole_corpus_index"].split("/")
    assert done == total and int(total) >= 2 * 256 and pm["value"] > 0 and pm["rows_per_shard"] % 16384 == 0
    assert len(pm["index_bytes_per_rank"]) == 2 and pm["remote_share_of_row_reads"] == 0.5


def test_bench_four_ranks_on_one_device(gpu):
    """world 4 rehearsed on one GPU (four rank processes): the replicas line, the sharded leg (host-staged exchange) and the
    peer-mapped leg (four dmabuf descriptors crossing a Unix socket; three quarters of the row reads are other ranks' rows),
    each parity-gated; the measured line was armed as the process's last words while the side legs ran and disarmed after."""
    j = _run("--gpus", "4", "--steps", "2", "--warmup", "1", "--rows", "400000", "--nq", "512", "--n-to-score", "1500",
             "--sharded-nq", "64", "--exchange", "host", "--single-device")
    _contract(j, 4, 2)
    assert j["value"] == j["replicas"]["value"] > 0
    done, total = j["sharded"]["parity_vs_single_gpu"].split("/")
    assert done == total and int(total) >= 4 * 64
    pm = j["peer_mapped"]
    assert pm.get("error") is None, pm
    done, total = pm["parity_vs_whole_corpus_index"].split("/")
    assert done == total and int(total) >= 4 * 256 and pm["remote_share_of_row_reads"] == 0.75
    assert len(pm["index_bytes_per_rank"]) == 4


def test_bench_under_the_drivers_launcher(gpu):
    """the driver starts N > 1 as `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P
    bench.py --gpus N ...`: bench.py takes RANK / WORLD_SIZE / MASTER_* from that environment (its own TCP star on MASTER_PORT + 1;
    torch itself is never imported by bench.py) and rank 0 prints the one line."""
    pytest.importorskip("torch")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", "29533", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                          "--rows", "300000", "--nq", "512", "--n-to-score", "1500", "--sharded-nq", "64", "--exchange", "host",
                          "--single-device"], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1
    j = json.loads(lines[0])
    _contract(j, 2, 2)
    assert j["value"] == j["replicas"]["value"] > 0
    done, total = j["sharded"]["parity_vs_single_gpu"].split("/")
    assert done == total and int(total) >= 2 * 64
    assert j["peer_mapped"].get("error") is None
