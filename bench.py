#!/usr/bin/env python3
"""bench.py -- BEVBert pre-training throughput on MI355X (BASELINE.json metric: pre-train samples/sec).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is one full optimisation step (lift+splat, forward, backward, gradient all-reduce, clip, AdamW) on one
synthetic R2R-shaped batch per GPU: BASELINE.json configs[1] -- scripts/pt_r2r.bash shapes, batch 64 per GPU,
autocast-style bf16 (fp32 masters), task mix mlm.5.sap.5.masksem.1, dropout 0.1 -- with inputs resident in HBM.
Weak scaling: the per-GPU batch is fixed, ranks draw different batches (seed + rank), value = global samples / s.

Rank 0 prints ONE JSON line.  Besides the contract fields it carries
  "roofline":     the dominant hand-written kernel, timed live with HIP events on the launching stream;
  "cpu_baseline": the CPU oracle (oracle/bevbert_ref.py, a restatement pinned to the reference) doing the same
                  fwd+bwd+AdamW on the host cores, on a bounded sample (rank 0, N = 1 only);
  "kernels":      per-kernel-class time of one profiled step of each task (custom kernels vs library GEMMs).
"""
import argparse
import json
import os
import sys
import time

os.environ.setdefault("TENSILE_STREAMK_DATA_PARALLEL", "1")     # see vln_bevbert_amd/__init__.py; before hipBLASLt loads

import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
MFMA_BF16_PEAK_TFLOPS = 2500.0  # dense bf16
MFMA_F32_PEAK_TFLOPS = 157.3    # v_mfma_f32_16x16x4_f32: the fp32 vector rate (MI355X_MICROARCH.md)


DEFAULT_RESIDUAL = "fp32"

T_START = time.perf_counter()


def log(msg):
    """progress to stderr (stdout carries only the JSON line)"""
    print(f"[bench +{time.perf_counter() - T_START:7.1f}s] {msg}", file=sys.stderr, flush=True)


def usable_cores():
    """cores this process may really use: affinity mask capped by the cgroup CPU quota (os.cpu_count() reports the
    whole host and oversubscribes a container)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, min(n, 64))


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=110, help="timed steps (default: ten 11-step task cycles)")
    ap.add_argument("--warmup", type=int, default=22)
    ap.add_argument("--batch", type=int, default=64, help="per-GPU batch (BASELINE.json configs[1]: 64)")
    ap.add_argument("--txt-len", type=int, default=80)
    ap.add_argument("--config", default="r2r", choices=["r2r", "rxr", "ce"],
                    help="r2r = BASELINE.json configs[1] (the bench line); rxr (xlm-roberta vocabulary, use --txt-len 160) "
                         "and ce (continuous-environment fork) are side measurements")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--residual", default=DEFAULT_RESIDUAL, choices=["fp32", "bf16"],
                    help="residual stream of the bf16 run: fp32 = torch.autocast's arithmetic (LayerNorm outputs and residual "
                         "sums of the post-norm blocks stay fp32 next to the bf16 copy the GEMMs read: train_r2r.py:256-258), "
                         "bf16 = every activation rounded to bf16 (twice the roundings of the reference's autocast run)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-pass", action="store_true")
    ap.add_argument("--cpu-threads", type=int, default=0)
    ap.add_argument("--save-gemm-tuning", default="", help="write the hipBLASLt choice table of this run to a file")
    ap.add_argument("--no-side", action="store_true",
                    help="skip the 'side_configs' block (RxR per-rank shape of BASELINE.json configs[3], the CE fork's model, the "
                         "fine-tune rollout of configs[4]; each runs as a short child process after the main measurement)")
    ap.add_argument("--no-fwd", action="store_true", help="skip the forward-only timing")
    ap.add_argument("--no-stream", action="store_true",
                    help="skip the 'sustained' block (live loader: host-side index building + refills of double-buffered "
                         "static batches on a copy stream, every step)")
    ap.add_argument("--ragged", action="store_true",
                    help="sustained block on ragged batches (T in [1,7], text length in [L/2, L], 36..38 views): many shape "
                         "buckets, reports buckets / captures / eager steps")
    ap.add_argument("--sustained-ragged", action=argparse.BooleanOptionalAction, default=True,
                    help="a SECOND sustained block ('sustained_ragged') on ragged batches, next to the fixed-shape one "
                         "(one GPU, in a child process; --no-sustained-ragged skips it)")
    ap.add_argument("--txt-len-min", type=int, default=0,
                    help="shortest instruction of the ragged sustained run (default: half of --txt-len)")
    ap.add_argument("--ship-grid", action="store_true",
                    help="sustained block ships the grid features of every batch over PCIe (462 MB fp32 at batch 64) instead "
                         "of reading rows of the device-resident feature store")
    ap.add_argument("--stream-steps", type=int, default=33)
    ap.add_argument("--loader-workers", type=int, default=4,
                    help="DataLoader worker processes that collate the sustained run's batches (0: in the producer thread)")
    ap.add_argument("--dry-launch", action="store_true",
                    help="only exercise the rank launch: every rank prints its RANK / WORLD_SIZE / LOCAL_RANK as one JSON line and "
                         "exits (no GPU needed; tests/test_host_logic.py)")
    ap.add_argument("--detail", default=os.path.join(ROOT, "bench_detail.json"),
                    help="where the full record goes (per-kernel tables, every CPU cell, full side configs); the stdout line is "
                         "the compact contract line")
    ap.add_argument("--launch", default="auto", choices=["auto", "eager", "graph"],
                    help="how the step is issued: eager (~700 launches from Python), graph (hipGraph replay), or auto = time "
                         "both in the untimed preparation and keep the faster one (BEVBERT_GRAPHS=0/1 in the environment "
                         "forces eager / graph as well)")
    return ap.parse_args()


def algorithmic_work(key, args, esize):
    """(flops, bytes) one launch of a traced C-ABI call is worth ALGORITHMICALLY."""
    if "[rows=" in key:                      # the row kernels are traced per problem size (ops_core.call)
        key = key.split("[")[0]
    if key.startswith("bevbert_attn_fwd"):
        B, nh, Lq, Lk = args[8], args[9], args[10], args[11]
        return 4.0 * B * nh * Lq * Lk * 64, (2 * Lq + 2 * Lk) * nh * 64 * B * esize
    if key.startswith("bevbert_attn_bwd"):
        B, nh, Lq, Lk = args[14], args[15], args[16], args[17]
        return 10.0 * B * nh * Lq * Lk * 64, (4 * Lq + 4 * Lk) * nh * 64 * B * esize
    if key == "bevbert_bias_dropout_residual_layernorm_fwd":
        rows, H = args[9], args[10]
        n_in = 1 + (args[2] is not None)
        return 0.0, rows * H * esize * (n_in + 1 + (args[6] is not None and args[6] != args[0]))
    if key == "bevbert_layernorm_bwd":
        rows, H = args[11], args[12]
        return 0.0, rows * H * esize * (2 + (args[5] is not None) + (args[6] is not None))
    if key in ("bevbert_bias_gelu_fwd", "bevbert_bias_relu_fwd"):
        return 0.0, args[3] * args[4] * esize * 2
    if key in ("bevbert_bias_gelu_bwd", "bevbert_bias_relu_bwd"):
        return 0.0, args[6] * args[7] * esize * 3
    if key == "bevbert_colsum":
        return 0.0, args[3] * args[4] * esize
    if key == "bevbert_bev_splat_mean":
        B, P, K, C = args[6], args[7], args[8], args[9]
        in_size = {0: 4, 1: 2, 2: 2}[args[1]]
        out_size = {0: 4, 1: 2, 2: 2}[args[5]]
        return 0.0, B * (P * C * in_size + P * 4 + K * C * out_size + P * 1 + K * 41)
    if key == "bevbert_adamw_step":
        return 0.0, args[7] * (16 + 12 + 2)
    if key == "bevbert_grad_norm_clip":
        return 0.0, args[1] * 4
    if key.startswith("gemm:"):
        m, n, k = args
        return 2.0 * m * n * k, (m * k + n * k + m * n) * esize
    size_of = {0: 4, 1: 2, 2: 2}       # dtype codes of the C ABI (f32, bf16, f16)
    if key == "bevbert_layernorm_bwd_add":          # dy, z, dz_add in; dz (and dx behind a dropout mask) out
        rows, H = args[12], args[13]
        return 0.0, rows * H * size_of[args[14]] * (3 + (args[5] is not None) + (args[6] is not None and args[6] != args[5]))
    if key == "bevbert_layernorm_post_fwd":         # x (+ post terms) in; y (and z) out
        rows, H = args[10], args[11]
        return 0.0, rows * H * size_of[args[13]] * (1 + (args[4] is not None) + (args[5] is not None) + 1 + (args[7] is not None))
    if key == "bevbert_embed_sum_layernorm_fwd":    # gathered word rows in; y (and z) out
        rows, H = args[10], args[12]
        return 0.0, rows * H * size_of[args[14]] * (2 + (args[7] is not None)) + rows * 8
    if key in ("bevbert_multi_accum", "bevbert_multi_finalize"):      # device task tables: bytes recorded when built
        from vln_bevbert_amd import ops as _ops
        return 0.0, float(_ops.ReduceQueue.table_bytes.get(args[0], 0))
    if key == "bevbert_attn_drop_bits":             # the keep-bit matrix (forward layout; + the key-major and the per-lane
        B, nh, Lq, Lk = args[1], args[2], args[3], args[4]   # layouts for 256 < Lk <= 448: attn_bwd3 / attn_fwd4)
        words = B * nh * ((Lq + 127) // 128 * 8) * ((Lk + 63) // 64) * 16
        return 0.0, 8.0 * words * (3 if 256 < Lk <= 448 else 1)
    if key == "bevbert_colsum_partials":
        return 0.0, args[2] * args[3] * size_of[args[4]]
    if key in ("bevbert_embedding_grad", "bevbert_embedding_grad_sliced"):
        rows, H = args[3], args[4]
        return 0.0, rows * H * size_of[args[6] if key == "bevbert_embedding_grad" else args[7]] + rows * 8
    if key == "bevbert_dropout_add":
        n = args[3]
        return 0.0, n * (size_of[args[4]] + size_of[args[5]] * (1 + (args[1] is not None)))
    if key == "bevbert_cross_entropy_fwd":
        return 0.0, args[4] * args[5] * size_of[args[6]]
    if key == "bevbert_cross_entropy_bwd":
        return 0.0, args[5] * args[6] * size_of[args[7]] * 2
    if key == "bevbert_cast_f32":
        return 0.0, args[2] * (4 + size_of[args[3]])
    if key == "bevbert_accum_partials":
        return 0.0, args[2] * args[3] * size_of[args[4]] + 8 * args[3]
    if key == "bevbert_layernorm_res32_fwd":        # x bf16 + residual (fp32 | bf16) in; y bf16, y fp32 (, z fp32) out
        rows, H = args[11], args[12]
        return 0.0, rows * H * (2 + size_of[args[3]] + 2 + 4 + (4 if args[8] is not None else 0))
    if key == "bevbert_layernorm_res32_bwd":        # dy bf16, dy fp32, z fp32 in; dz (fp32 | bf16), dx bf16 out
        rows, H = args[12], args[13]
        return 0.0, rows * H * ((2 if args[0] is not None else 0) + (4 if args[1] is not None else 0) + 4
                                + (size_of[args[18]] if args[6] is not None else 0) + (2 if args[7] is not None else 0))
    if key == "bevbert_smallk_linear_layernorm_fwd":    # feat in; post1, gathered table row in; y out (no z)
        rows, K, H = args[11], args[12], args[13]
        e = size_of[args[15]]
        return 0.0, rows * (K * 4 + H * e * (1 + (args[5] is not None) + (args[6] is not None)))
    if key == "bevbert_smallk_linear_layernorm_bwd":    # dy and feat in; nothing but the parameter gradients out
        rows, K, H = args[12], args[13], args[14]
        return 0.0, rows * (K * 4 + H * size_of[args[15]])
    if key == "bevbert_rows_gather":
        return 0.0, args[3] * args[4] * size_of[args[5]] * 2 + args[3] * 8
    if key == "bevbert_rows_scatter":
        return 0.0, args[3] * args[4] * size_of[args[5]] * 2 + args[3] * 8
    if key == "bevbert_zero":
        return 0.0, float(args[1])
    if key == "bevbert_colsum_any":
        return 0.0, args[2] * args[3] * size_of[args[4]]
    if key in ("bevbert_graph_bias_fwd",):
        return 0.0, args[4] * 8.0
    if key == "bevbert_graph_bias_bwd":
        return 0.0, args[2] * args[3] * args[4] * args[5] * args[5] * 4.0
    if key in ("bevbert_weighted_mean_fwd", "bevbert_weighted_mean_bwd"):
        return 0.0, args[4] * 8.0
    if key == "bevbert_bev_lift_bin":               # depths + poses in; cell ids, sorted order, cell starts out
        B, V, hw, dim = args[5], args[6], args[7], args[9]
        return 0.0, B * (V * hw * hw * (4 + 4 + 4) + V * 64 + (dim * dim + 1) * 4)
    if key == "bevbert_segment_wsum":               # lower bound: the edge count lives in the device CSR
        return 0.0, args[5] * args[6] * size_of[args[7]] * 2
    return 0.0, 0.0


def spawn_ranks(a):
    """`python bench.py --gpus N` without a launcher: become `torch.distributed.run` with N ranks of this script on this
    node (the reference starts its ranks the same way, scripts/pt_r2r.bash:8-10 `python -m torch.distributed.launch
    --nproc_per_node`).  The ranks inherit stdout / stderr; rank 0 prints the line."""
    import socket
    if not a.dry_launch and torch.cuda.device_count() < a.gpus and os.environ.get("BEVBERT_BENCH_SHARE_GPU") != "1":
        sys.exit(f"bench.py: --gpus {a.gpus}, but {torch.cuda.device_count()} GPUs are visible")
    with socket.socket() as sk:             # a free port for the rendezvous (two benches on one node must not collide)
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("OMP_NUM_THREADS", str(max(1, usable_cores() // a.gpus)))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC: RCCL's only working transport on these hosts
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    log(f"starting {a.gpus} ranks: {' '.join(cmd[1:9])} bench.py ...")
    sys.stdout.flush()
    sys.stderr.flush()
    os.execve(sys.executable, cmd, env)


LINE_LIMIT = 6144          # bytes: the driver parses the LAST stdout line; round 5's 24 KB line was not parsed


def _short_roofline(blk):
    if not isinstance(blk, dict):
        return blk
    keep = ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "avg_launch_us", "launches", "entry_frac",
            "frac_back_to_back")
    return {k: blk[k] for k in keep if k in blk}


def compact_line(out, detail_path=None):
    """The contract line: what the driver parses (contract fields, config, roofline, cpu_baseline) plus one short figure
    per optional block.  Everything else -- per-kernel tables, per-shape rooflines, all CPU cells, the full side configs,
    the RCCL digest -- lives in the detail record (``--detail``, default bench_detail.json next to this file)."""
    first = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
             "vs_baseline", "dtype", "data", "config", "step_launch", "launch_calibration", "graph_error", "final_loss",
             "host_enqueue_ms_per_step", "whole_cycles", "fwd_ms_per_batch")
    line = {k: out[k] for k in first if k in out}
    for k in ("roofline", "roofline_attn_bwd", "roofline_attn_fwd", "roofline_attn_short"):
        if k in out:
            line[k] = _short_roofline(out[k])
    cb = out.get("cpu_baseline")
    if isinstance(cb, dict):
        c = {k: cb[k] for k in ("value", "unit", "cores", "kind", "sample", "error") if k in cb}
        fw = cb.get("forward") or {}
        if fw:
            c["host_cpu"] = fw.get("host_cpu")
            # BASELINE.json configs[0]: the batch-2 forward on all usable cores, SAP and MLM
            c["forward_b2_all_cores"] = {x["task"]: {"median_s": x["median_s"], "samples_per_s": x["samples_per_s"]}
                                         for x in fw.get("cells", []) if x["batch"] == 2 and x["threads"] == fw.get("usable_cores")}
        line["cpu_baseline"] = c
    for k in ("sustained", "sustained_ragged"):
        v = out.get(k)
        if isinstance(v, dict):
            line[k] = {x: v[x] for x in ("samples_per_s", "ms_per_step", "vs_resident", "steps_eager", "buckets", "error") if x in v}
    if isinstance(out.get("kernels"), dict):
        kn = out["kernels"]
        line["kernels"] = {x: kn[x] for x in ("profiled_steps", "wall_ms", "custom_kernel_ms", "library_gemm_ms",
                                              "library_gemm_tflops", "other_ms", "custom_ms_share_with_roofline") if x in kn}
    sc = out.get("side_configs")
    if isinstance(sc, dict):
        line["side_configs"] = {}
        for name, d in sc.items():
            if not isinstance(d, dict):
                continue
            short = {x: d[x] for x in ("value", "ms_per_step", "ms_per_nav_step", "episodes_per_s") if x in d}
            if "error" in d or "skipped" in d:
                short["note"] = str(d.get("error") or d.get("skipped"))[:80]
            line["side_configs"][name] = short
    rc = out.get("rccl")
    if isinstance(rc, dict):
        line["rccl"] = {x: rc[x] for x in ("ranks", "backend", "exchange", "collectives_wait_behind_compute", "allreduce_bytes_per_step", "allreduce_alone_ms",
                                          "allreduce_alone_GBps", "eager_ms_per_step_by_exchange", "first_collective_at_fraction_of_backward",
                                          "error") if x in rc}
    if detail_path:
        line["detail"] = os.path.basename(detail_path)
    # the limit is a hard property of the line: shed optional blocks (least important first) rather than exceed it
    for drop in ("side_configs", "kernels", "sustained_ragged", "sustained", "fwd_ms_per_batch", "rccl", "roofline_attn_short",
                 "launch_calibration", "whole_cycles"):
        if len(json.dumps(line)) < LINE_LIMIT:
            break
        line.pop(drop, None)
    return line


def main():
    a = parse()
    # watchdog: a stalled run (a collective that never completes, a wedged queue) dumps every thread's Python stack to
    # stderr and exits instead of sitting on the GPU until somebody kills it; a normal run takes 1-3 minutes
    import faulthandler
    faulthandler.dump_traceback_later(int(os.environ.get("BEVBERT_BENCH_WATCHDOG_S", "1500")), exit=True)
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(a)                      # does not return: this process becomes the launcher of --gpus ranks
    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    if world != a.gpus:
        sys.exit(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={world}: pass the launcher's --nproc-per-node as --gpus "
                 "(or run plain `python bench.py --gpus N`, which starts its own ranks)")
    if a.dry_launch:
        print(json.dumps({"dry_launch": True, "rank": rank, "world_size": world, "local_rank": local_rank,
                          "master": f"{os.environ.get('MASTER_ADDR', '')}:{os.environ.get('MASTER_PORT', '')}"}), flush=True)
        return
    if not torch.cuda.is_available():
        sys.exit("bench.py measures the MI355X path; no GPU is visible")
    # BEVBERT_BENCH_SHARE_GPU=1: a REHEARSAL of the multi-rank code path on a one-GPU box -- every rank uses GPU 0 and the
    # ranks exchange over gloo (RCCL refuses two ranks on one device).  Not a measurement: the line says so in "data".
    share_gpu = world > 1 and os.environ.get("BEVBERT_BENCH_SHARE_GPU") == "1"
    if share_gpu:
        local_rank = 0
    if torch.cuda.device_count() <= local_rank:
        sys.exit(f"bench.py: rank {rank} wants GPU {local_rank}, but only {torch.cuda.device_count()} are visible")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # BEVBERT_FORCE_COLLECTIVES=1: run the RCCL exchange (side stream, backward hook, in-place all-reduce) on a one-rank
    # group too -- a single-GPU stress of the multi-GPU code path, not a benchmark configuration
    force = os.environ.get("BEVBERT_FORCE_COLLECTIVES") == "1"
    nccl_log = None
    if world > 1 or force:
        import torch.distributed as dist
        # what RCCL decides for this topology (rings / trees, channels, protocol) goes into a per-rank file that rank 0
        # digests into the "rccl" block: the first multi-GPU run of this code must be readable without a second run
        nccl_log = f"/tmp/bevbert_rccl_{os.getpid()}_r{rank}.log"
        os.environ.setdefault("NCCL_DEBUG", "INFO")
        os.environ.setdefault("NCCL_DEBUG_SUBSYS", "INIT,GRAPH,TUNING")
        os.environ.setdefault("NCCL_DEBUG_FILE", nccl_log)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        if share_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from vln_bevbert_amd import lib, ops, synthetic
    from vln_bevbert_amd.static_step import StaticBatch
    from vln_bevbert_amd.config import BevBertConfig
    from vln_bevbert_amd.pretrain_cmt import GlocalTextPathCMTPreTraining
    from vln_bevbert_amd.train import PretrainTrainer, load_gemm_tuning
    n_tuned = load_gemm_tuning()
    n_rows = ops.load_gemm_tuning_table()
    log(f"hipBLASLt choice table: {n_rows} rows ({ops.GEMM_TUNING_FILE}); torch TunableOp table: {n_tuned} shapes")

    log(f"rank {rank}/{world} on {torch.cuda.get_device_name(dev)}; building model")
    cdt = torch.bfloat16 if a.dtype == "bf16" else torch.float32
    esize = 2 if a.dtype == "bf16" else 4
    cfg = {"r2r": BevBertConfig, "rxr": BevBertConfig.rxr, "ce": BevBertConfig.ce}[a.config]()   # configs/*_model.json
    torch.manual_seed(0)                                    # identical initial weights on every rank
    model = GlocalTextPathCMTPreTraining(cfg)
    arena = model.finalize(dev, cdt, torch.float32 if (a.residual == "fp32" and a.dtype == "bf16") else None)
    model.train()
    model.set_dropout(0.1)                                  # train_r2r.py:157
    trainer = PretrainTrainer(model, arena, rank=rank, world_size=world, force_collectives=force)
    launch = a.launch
    if launch == "auto" and os.environ.get("BEVBERT_GRAPHS") in ("0", "1"):
        launch = "graph" if os.environ["BEVBERT_GRAPHS"] == "1" else "eager"
    # Multi-rank runs take the same path as one rank: the step is captured with the in-place RCCL all-reduce of the
    # gradient arena inside (issued from the backward hooks on the reducer's stream, forked from and joined to the
    # capturing stream), eager and replayed cycles are timed in the untimed preparation (maximum over ranks, so every
    # rank takes the same decision) and the faster mechanism runs the timed region.  A capture that fails -- a collective
    # library that refuses stream capture -- is reported in "graph_error" and the trainer continues eagerly (train.py);
    # the eager step costs 16-21 ms of host time per step on top of nothing the GPU could not hide at 18 ms/step.
    trainer.use_graphs = launch != "eager"
    # the reference draws the task of each step at random with ratio 5:5:1 (MetaLoader); the bench walks that mix as
    # a fixed 11-step cycle so that every run (and every K that is a multiple of 11) times exactly the same work
    cycle = ["mlm", "sap", "mlm", "sap", "mlm", "sap", "masksem", "mlm", "sap", "mlm", "sap"]
    cycle = [t if t in cfg.pretrain_tasks else "mlm" for t in cycle]      # the CE fork trains mlm + sap only
    tasks = tuple(dict.fromkeys(cycle))
    counter = [0]

    log(f"model in arena: {arena.n_params / 1e6:.1f} M params; generating resident batches")
    # resident synthetic batches: two per task and rank, drawn with seed 1000 + rank, each in
    # its own set of static device buffers (static_step.StaticBatch: loader-built index tensors, padded row counts)
    batches = {t: [StaticBatch(cfg, t, synthetic.make_batch(cfg, t, a.batch, seed=1000 + rank + 97 * j,
                                                            txt_len=a.txt_len, sems_as="ids"), dev)
                   for j in range(2)] for t in tasks}
    torch.cuda.synchronize()

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def run(n):
        losses = []
        for _ in range(n):
            i = counter[0]
            counter[0] += 1
            t = cycle[i % len(cycle)]
            losses.append(trainer.step(t, batches[t][(i // len(cycle)) % 2]))
        return losses

    # preparation (untimed, before the warm-up): every distinct resident batch goes through the step eagerly (each GEMM
    # problem of the run gets its hipBLASLt plan -- from the shipped choice table, or timed now: problems whose row
    # count depends on the data) and is then captured into a hipGraph, so that the timed region only replays
    n_prep = trainer.GRAPH_WARMUP + 1 if trainer.use_graphs else 1
    log(f"batches resident; preparation pass ({n_prep} steps per batch; graphs {'on' if trainer.use_graphs else 'off'})")
    for _ in range(n_prep):
        for t in tasks:
            for bt in batches[t]:
                trainer.step(t, bt)
    n_graphs = sum(bt.graph is not None for t in tasks for bt in batches[t])
    calibration = None
    if launch == "auto":
        # Eager issue and graph replay execute the same kernels on the same buffers (bit-identical losses:
        # tests/test_gpu_zz_streams.py); which one is faster depends on the box: eager keeps the weight-gradient work on
        # a second hardware queue but needs ~16-21 ms of host time per step; the replay needs ~2 ms and overlaps more
        # streams, but runs through ROCm's multi-stream graph path.  Time one task cycle of each (untimed preparation)
        # and keep the faster mechanism for the warm-up and the timed region.
        def cycle_ms(use_graphs):
            trainer.use_graphs = use_graphs
            run(len(cycle))
            barrier()
            t0 = time.perf_counter()
            run(len(cycle))
            barrier()
            return 1000.0 * (time.perf_counter() - t0) / len(cycle)
        ms = torch.tensor([cycle_ms(False), cycle_ms(True)], device=dev, dtype=torch.float64)
        if world > 1:
            dist.all_reduce(ms, op=dist.ReduceOp.MAX)       # every rank takes the same decision
        ms_eager, ms_graph = (float(x) for x in ms.tolist())
        trainer.use_graphs = ms_graph < ms_eager
        calibration = {"eager_ms_per_step": round(ms_eager, 3), "graph_ms_per_step": round(ms_graph, 3)}
        log(f"launch calibration: eager {ms_eager:.2f} ms/step, hipGraph replay {ms_graph:.2f} ms/step -> "
            f"{'graph' if trainer.use_graphs else 'eager'}")
    # host cost of issuing ONE step with nothing queued behind it: a burst of 6 steps after a device synchronise.  The
    # timed region's host_enqueue figure also contains the time the runtime makes the host wait once ~10 graph launches
    # are in flight (back-pressure, not work: with 11 timed steps it reads 1.9 ms/step, with 22 steps 9.7 ms/step).
    barrier()
    t0 = time.perf_counter()
    run(6)
    host_burst_ms = 1000.0 * (time.perf_counter() - t0) / 6
    barrier()
    log("warm-up")
    run(a.warmup)
    barrier()
    log(f"warm-up done; timing {a.steps} steps")
    t0 = time.perf_counter()
    losses = run(a.steps)
    t_host = time.perf_counter() - t0          # host time to ENQUEUE the steps (the GPU is still running)
    barrier()
    dt = time.perf_counter() - t0
    log(f"timed region: {1000 * dt / a.steps:.2f} ms/step (host enqueue {1000 * t_host / a.steps:.2f} ms/step)")
    if world > 1:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    value = a.steps * a.batch * world / dt
    resident_events = None
    if os.environ.get("BEVBERT_STEP_EVENTS") == "1":        # diagnosis (after the timed region): device time per step + gaps
        marks = []
        for _ in range(3 * len(cycle)):
            t = cycle[counter[0] % len(cycle)]
            marks.append((t, torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)))
            marks[-1][1].record()
            run(1)
            marks[-1][2].record()
        barrier()
        resident_events = _step_event_digest(marks)
        log(f"step events, resident: {resident_events}")
    whole = None
    if a.steps % len(cycle):
        # K is not a whole number of 11-step task cycles (the driver's --steps 20): the K steps above are the contract's
        # timed region; a second region of ceil(K / 11) whole cycles, continuing the same walk from a cycle boundary, gives
        # the figure for the exact 5:5:1 task mix next to it
        run((-counter[0]) % len(cycle))
        n_whole = -(-a.steps // len(cycle)) * len(cycle)
        barrier()
        t0 = time.perf_counter()
        run(n_whole)
        barrier()
        dtw = time.perf_counter() - t0
        if world > 1:
            tt = torch.tensor([dtw], device=dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dtw = float(tt.item())
        whole = {"steps": n_whole, "ms_per_step": round(1000.0 * dtw / n_whole, 3),
                 "value": round(n_whole * a.batch * world / dtw, 2)}
        log(f"whole task cycles: {n_whole} steps at {whole['ms_per_step']:.2f} ms/step")

    out = {
        "metric": "pretrain_samples_per_sec", "value": round(value, 2), "unit": "samples/s", "n_gpus": world,
        "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(1000.0 * dt / a.steps, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": a.dtype + ("+fp32-residual" if a.residual == "fp32" and a.dtype == "bf16" else ""),
        "data": "synthetic" if not share_gpu else "synthetic; REHEARSAL: all ranks share GPU 0 and exchange over gloo -- not a measurement",
        "config": {"workload": f"{a.config.upper()} pre-train step (lift+splat, fwd, bwd, all-reduce, clip, AdamW), "
                               "scripts/pt_r2r.bash shapes: 36 views x 512, 5-step paths, 2352 grid points x 768 -> "
                               f"{cfg.bev_dim}x{cfg.bev_dim} BEV, {a.txt_len}-token text, task cycle "
                               f"{'.'.join(f'{t}.{cycle.count(t)}' for t in tasks)} (fixed 11-step cycle), dropout 0.1",
                   "batch_per_gpu": a.batch, "global_batch": a.batch * world, "parallelism": f"dp{world}",
                   "params_M": round(arena.n_params / 1e6, 1), "gemm": f"hipBLASLt via the C ABI, {n_rows} shapes from the shipped choice table, others timed on first use"},
        "host_enqueue_ms_per_step": round(1000.0 * t_host / a.steps, 3),
        "host_issue_ms_per_step_unthrottled": round(host_burst_ms, 3),
        "step_launch": f"hipGraph replay ({n_graphs} captured steps, one per resident batch)"
                       if (n_graphs and trainer.use_graphs) else "eager",
        "launch_calibration": calibration,
        "graph_error": trainer.graph_error,
        # library GEMM algorithms dropped because two launches on the same operands gave different bits (gemm.hip)
        "gemm_plans": ops.gemm_plan_count(), "gemm_candidates_rejected_as_not_reproducible": lib.load().bevbert_gemm_rejected_count(),
        "final_loss": round(float(losses[-1].item()), 4),
    }
    if whole is not None:
        out["whole_cycles"] = whole
    if resident_events is not None:
        out["step_events_resident"] = resident_events

    # ---- forward ms/batch (the second half of BASELINE.json's metric; reference: train_r2r.py:256-260): the training
    # forward (dropout on, tape recorded) issued eagerly, and the same batch's inference forward replayed from a graph
    if rank == 0 and not a.no_fwd:
        log("forward timing")
        fwd = {}
        for t in tasks:
            sb = batches[t][0]
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]

            def timed(fn, n=10):
                for _ in range(2):
                    fn()
                torch.cuda.synchronize()
                ev[0].record()
                for _ in range(n):
                    fn()
                ev[1].record()
                torch.cuda.synchronize()
                return ev[0].elapsed_time(ev[1]) / n

            def train_fwd():
                ops.RT.new_step(12345)
                model.loss_mean(sb.tensors, t)

            fwd[t] = {"train_eager_ms": round(timed(train_fwd), 3)}
            arena.sync()
            model.eval()
            try:
                with torch.no_grad():
                    for _ in range(2):
                        model.loss_mean(sb.tensors, t)
                    torch.cuda.synchronize()
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g, capture_error_mode="thread_local"):
                        model.loss_mean(sb.tensors, t)
                    fwd[t]["eval_graph_ms"] = round(timed(g.replay), 3)
            except Exception as e:          # a forward that cannot be captured is reported, not hidden
                fwd[t]["eval_graph_ms"] = None
                fwd[t]["eval_graph_error"] = repr(e)[:200]
                lib.load().bevbert_hip_error_reset()
                torch.cuda.synchronize()
            model.train()
        out["fwd_ms_per_batch"] = fwd
    if rank == 0 and a.no_fwd:
        pass
    if world > 1 or force:
        # what the exchange moves, and how long it takes alone (all ranks take part; rank 0 reports)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            dist.all_reduce(arena.grads)
        torch.cuda.synchronize()
        ar_ms = (time.perf_counter() - t0) / 5 * 1e3
        # one eager step per task with per-region events: when each arena region's all-reduce starts and ends relative to
        # the start of the step and to the end of backward
        regions = {}
        was_graphs = trainer.use_graphs
        trainer.use_graphs = False
        for t in tasks:
            trainer.reducer.timeline = []
            s_ev, b_ev = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            s_ev.record()
            trainer.forward_backward(t, batches[t][0])
            b_ev.record()
            torch.cuda.synchronize()
            regions[t] = {"forward_backward_ms": round(s_ev.elapsed_time(b_ev), 3),
                          "regions": [{"MB": round((hi - lo) * 4 / 1e6, 1), "start_ms": round(s_ev.elapsed_time(e0), 3),
                                       "end_ms": round(s_ev.elapsed_time(e1), 3)} for lo, hi, e0, e1 in trainer.reducer.timeline]}
            bs = getattr(trainer, "backward_start", None)
            if bs is not None and regions[t]["regions"]:
                # where in the backward pass the first collective is enqueued (before 30 % of it)
                b0, fb = s_ev.elapsed_time(bs), regions[t]["forward_backward_ms"]
                regions[t]["backward_start_ms"] = round(b0, 3)
                regions[t]["first_collective_at_fraction_of_backward"] = round(
                    (regions[t]["regions"][0]["start_ms"] - b0) / max(fb - b0, 1e-6), 3)
            trainer.reducer.timeline = None
        # the same eager step with each wire format of the exchange (BEVBERT_GRAD_EXCHANGE): fp32 in-place all-reduce (the
        # reference's DDP semantics, the bench line) against bf16 reduce-scatter + all-gather (half the bytes per xGMI link);
        # one task cycle each, issued eagerly in both cases so that the two figures differ by the exchange only
        by_exchange = {}
        ex0 = trainer.reducer.exchange
        for ex in ("fp32", "bf16"):
            try:
                trainer.reducer.exchange = ex
                run(2)
                barrier()
                t0 = time.perf_counter()
                run(len(cycle))
                barrier()
                tt = torch.tensor([1000.0 * (time.perf_counter() - t0) / len(cycle)], device=dev, dtype=torch.float64)
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                by_exchange[ex] = round(float(tt.item()), 3)
            except Exception as e:      # noqa: BLE001
                by_exchange[ex] = f"{type(e).__name__}: {e}"[:160]
                break
        trainer.reducer.exchange = ex0
        trainer.use_graphs = was_graphs
        fracs = [r.get("first_collective_at_fraction_of_backward") for r in regions.values()]
        digest = []
        try:
            import glob
            cands = sorted(glob.glob(os.environ.get("NCCL_DEBUG_FILE", nccl_log or "") + "*"))
            with open(cands[0]) as f:
                for ln in f:
                    if any(k in ln for k in ("Channel", "Ring", "Tree", "Algo", "Proto", "nChannels", "threshold", "XGMI", "NVL", "Connected")):
                        digest.append(ln.strip()[-160:])
        except Exception as e:      # noqa: BLE001
            digest = [f"no RCCL debug file: {e!r}"[:160]]
        qr = trainer.reducer.queue_report or {}
        behind = (qr.get("own_group") or qr).get("waits_behind_compute")
        out["rccl"] = {"ranks": dist.get_world_size(), "backend": dist.get_backend(),
                       "exchange": trainer.reducer.exchange, "region_timeline": regions,
                       # False = the collective stream has a hardware queue other than the compute stream's (hwqueues.py)
                       "collectives_wait_behind_compute": behind, "collective_queue_report": qr,
                       "debug_digest": digest[:40], "debug_lines": len(digest),
                       "allreduce_bytes_per_step": int(arena.numel) * 4,
                       "phase_a_bytes": int(arena.numel - trainer.reducer.split) * 4,
                       "allreduce_alone_ms": round(ar_ms, 3),
                       "allreduce_alone_GBps": round(arena.numel * 4 / ar_ms / 1e6, 1),
                       "eager_ms_per_step_by_exchange": by_exchange,
                       "first_collective_at_fraction_of_backward": [f for f in fracs if f is not None],
                       "overlap": "phase A from the text-embedding gradient hook, phase B after backward; both on a "
                                  "side stream" + (", captured inside the step graph" if n_graphs else "")}
        arena.grads.zero_()

    if not a.no_kernel_pass:
        # ---- per-kernel timing pass (not part of the timed region above): HIP events around every C-ABI launch.
        # Every rank runs the three steps (they contain the gradient all-reduce); only rank 0 records.
        log("kernel timing pass")
        if rank == 0:
            ops.RT.trace = {}
        prof_start = torch.cuda.Event(enable_timing=True)
        prof_end = torch.cuda.Event(enable_timing=True)
        prof_start.record()
        for t in tasks:
            trainer.step(t, batches[t][0])
        prof_end.record()
        torch.cuda.synchronize()
    if rank == 0 and not a.no_kernel_pass:
        trace, ops.RT.trace = ops.RT.trace, None
        total_ms = prof_start.elapsed_time(prof_end)
        rows = {}
        for key, evs in trace.items():
            ms = [s.elapsed_time(e) for s, e, _ in evs]
            fl = by = 0.0
            for _, _, args in evs:
                f, b = algorithmic_work(key, args, esize)
                fl += f
                by += b
            rows[key] = {"launches": len(ms), "ms": round(sum(ms), 3), "avg_us": round(1000 * sum(ms) / len(ms), 2),
                         "gflop": round(fl / 1e9, 2), "mb": round(by / 1e6, 2)}
        gemm_rows = {k: v for k, v in rows.items() if k.startswith("gemm:")}
        rows = {k: v for k, v in rows.items() if not k.startswith("gemm:")}
        custom_ms = sum(r["ms"] for r in rows.values())
        gemm_ms = sum(r["ms"] for r in gemm_rows.values())
        gemm_gflop = sum(r["gflop"] for r in gemm_rows.values())
        for r in gemm_rows.values():
            r["tflops"] = round(r["gflop"] / r["ms"], 1) if r["ms"] > 0 else 0.0
        out["kernels"] = {"profiled_steps": " + ".join(f"1 x {t}" for t in tasks), "wall_ms": round(total_ms, 2),
                          "custom_kernel_ms": round(custom_ms, 2),
                          "library_gemm_ms": round(gemm_ms, 2),
                          "library_gemm_tflops": round(gemm_gflop / max(gemm_ms, 1e-9), 1),
                          "other_ms": round(total_ms - custom_ms - gemm_ms, 2),
                          "by_kernel": dict(sorted(rows.items(), key=lambda kv: -kv[1]["ms"])[:40]),
                          "by_gemm": dict(sorted(gemm_rows.items(), key=lambda kv: -kv[1]["ms"])[:24])}
        # dominant hand-written kernel: the C-ABI entry with the largest total time over the profiled steps (all its
        # shapes together); the roofline is quoted for that entry's heaviest shape, so that "per launch" means one
        # problem size (and matches one row of the rocprofv3 summary under profiles/)
        by_entry = {}
        for k, r in rows.items():
            by_entry[k.split("[")[0]] = by_entry.get(k.split("[")[0], 0.0) + r["ms"]
        dom_entry = max(by_entry, key=by_entry.get)

        mfma_peak = MFMA_BF16_PEAK_TFLOPS if a.dtype == "bf16" else MFMA_F32_PEAK_TFLOPS

        def roof(r):
            secs = r["ms"] / 1e3
            if r["gflop"] > 0:
                ach = r["gflop"] / 1e3 / secs
                return {"bound": "mfma", "achieved": round(ach, 2), "peak": mfma_peak, "unit": "TFLOP/s",
                        "frac": round(ach / mfma_peak, 4)}
            ach = r["mb"] / 1e3 / secs
            return {"bound": "hbm", "achieved": round(ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(ach / HBM_PEAK_GBS, 4)}

        def entry_block(entry):
            """roofline of one C-ABI entry: quoted for its heaviest shape (one problem size = one row of the rocprofv3
            per-shape summary under profiles/), plus the entry over ALL its shapes of the profiled steps (entry_frac: total
            algorithmic work / total time; the short-sequence launches run far below the heaviest shape)"""
            cand = [(k, r) for k, r in rows.items() if k.split("[")[0] == entry]
            # heaviest shape = most algorithmic work per launch (the 441 x 441 problem of the attention entries; by total
            # time the many short text launches of the forward would win)
            key, top = max(cand, key=lambda kv: (max(kv[1]["gflop"], kv[1]["mb"] * 1e-3) / max(kv[1]["launches"], 1), kv[1]["ms"]))
            traffic = traffic_source = None      # HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/)
            for name in ("r06_pmc_traffic.json", "r05_pmc_traffic.json", "r04_pmc_traffic.json", "r03_pmc_traffic.json", "r02_pmc_traffic.json", "r01_pmc_traffic.json"):
                try:
                    with open(os.path.join(ROOT, "profiles", name)) as f:
                        traffic = json.load(f).get(key)
                except Exception:
                    traffic = None
                if traffic is not None:
                    traffic_source = (f"profiles/{name} (separate rocprofv3 --pmc passes over scripts/bench_attn_shape.py / "
                                      "bench_rowops.py, not this process)")
                    break
            ent_rows = [r for k, r in rows.items() if k.split("[")[0] == entry]
            ent = {"ms": sum(r["ms"] for r in ent_rows), "gflop": sum(r["gflop"] for r in ent_rows),
                   "mb": sum(r["mb"] for r in ent_rows)}
            blk = {"kernel": key, **roof(top), "traffic": traffic, "traffic_source": traffic_source,
                   "avg_launch_us": top["avg_us"], "launches": top["launches"],
                   "entry_share_of_custom_ms": round(by_entry[entry] / max(custom_ms, 1e-9), 3),
                   "entry_frac": roof(ent)["frac"], "entry_achieved": roof(ent)["achieved"],
                   "entry_launches": sum(r["launches"] for r in ent_rows),
                   "entry_shapes": {k.split("[", 1)[1].rstrip("]") if "[" in k else "all":
                                    {"launches": r["launches"], "avg_us": r["avg_us"], "frac": roof(r)["frac"]}
                                    for k, r in rows.items() if k.split("[")[0] == entry}}
            # Cross-check of the bracketed figure: the SAME C-ABI call (same arguments; its operands were activations of the
            # profiled step, whose memory is still mapped and no longer in use) ten times back to back between ONE pair of
            # events.  If the per-launch brackets contained launch latency the two would differ; on the MI355X they agree
            # within 0.1 % (239.2 vs 239.1 us), i.e. the bracket measures the kernel.  Only entries that are pure functions
            # of their operands are repeated (the optimiser updates state).
            if entry in ("bevbert_attn_fwd", "bevbert_attn_bwd"):
                try:
                    args0 = trace[key][0][2]
                    torch.cuda.synchronize()
                    s_ev, e_ev = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    ops._raw_call(entry, *args0)
                    s_ev.record()
                    for _ in range(10):
                        ops._raw_call(entry, *args0)
                    e_ev.record()
                    torch.cuda.synchronize()
                    b2b_us = 100.0 * s_ev.elapsed_time(e_ev)
                    f1, b1 = algorithmic_work(key, args0, esize)
                    unit_work = f1 / 1e12 if f1 > 0 else b1 / 1e9
                    peak = mfma_peak if f1 > 0 else HBM_PEAK_GBS
                    blk["back_to_back_us"] = round(b2b_us, 2)
                    blk["achieved_back_to_back"] = round(unit_work / (b2b_us * 1e-6), 2)
                    blk["frac_back_to_back"] = round(unit_work / (b2b_us * 1e-6) / peak, 4)
                except Exception as e:      # noqa: BLE001 -- an extra figure must not cost the bench line
                    blk["back_to_back_error"] = repr(e)[:200]
            return blk

        out["roofline"] = entry_block(dom_entry)
        # the hand-written MFMA kernels (rounds 1-4 quoted the 441 x 441 attention backward as the dominant kernel; since
        # round 5 the AdamW entry takes more time per step than all attention-backward shapes together): always reported
        for e_ in ("bevbert_attn_bwd", "bevbert_attn_fwd"):
            if e_ in by_entry and e_ != dom_entry:
                out["roofline_" + e_.split("_", 1)[1]] = entry_block(e_)
        # the same figure for every traced hand-written entry (heaviest first) -- context for the line above
        out["kernels"]["roofline_by_kernel"] = {
            k: {**roof(r), "avg_launch_us": r["avg_us"]} for k, r in sorted(rows.items(), key=lambda kv: -kv[1]["ms"])[:40]
            if r["gflop"] > 0 or r["mb"] > 0}
        # share of the hand-written kernel time whose entry has an algorithmic byte / flop count (>= 95 %)
        out["kernels"]["custom_ms_share_with_roofline"] = round(
            sum(r["ms"] for r in rows.values() if r["gflop"] > 0 or r["mb"] > 0) / max(custom_ms, 1e-9), 3)

    # the optional blocks come after the contract fields (value, roofline) and may fail without costing the line
    if world == 1 and not a.no_stream:
        log("sustained throughput with a live loader")
        try:
            out["sustained"] = sustained(cfg, a, trainer, cycle, tasks, dev, out["ms_per_step"])
        except Exception as e:      # noqa: BLE001 -- worker processes / shared memory / a loader thread: report, do not die
            out["sustained"] = {"error": f"{type(e).__name__}: {e}"[:300]}
            log(f"sustained block failed: {out['sustained']['error']}")
        if a.sustained_ragged and not a.ragged:
            # in a child process: a second loader / bucket manager in THIS process would garbage-collect the first one's
            # captured graphs while a capture of its own is open (hipErrorStreamCaptureUnsupported, r05)
            log("sustained throughput with a live loader, ragged batches (child process)")
            try:
                import subprocess
                cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--ragged", "--steps", "22", "--warmup", "11", "--no-side",
                       "--no-cpu-baseline", "--no-kernel-pass", "--no-fwd", "--config", a.config, "--batch", str(a.batch),
                       "--txt-len", str(a.txt_len), "--txt-len-min", str(a.txt_len_min), "--stream-steps", str(a.stream_steps),
                       "--dtype", a.dtype, "--residual", a.residual]
                child_detail = os.path.splitext(a.detail)[0] + "_ragged_child.json"
                pr = subprocess.run(cmd + ["--detail", child_detail], capture_output=True, text=True, timeout=600)
                d = json.loads([ln for ln in pr.stdout.strip().splitlines() if ln.startswith("{")][-1])
                out["sustained_ragged"] = d.get("sustained")
                try:                # the full block (loader timings, captures, waits) from the child's detail record
                    with open(child_detail) as f:
                        full = json.load(f)
                    if isinstance(full.get("sustained"), dict):
                        out["sustained_ragged"] = dict(full["sustained"], resident_ms_per_step_of_the_child=full.get("ms_per_step"))
                    os.remove(child_detail)
                except (OSError, ValueError):
                    pass
            except Exception as e:      # noqa: BLE001
                out["sustained_ragged"] = {"error": f"{type(e).__name__}: {e}"[:300]}
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        log("cpu baseline (oracle)")
        try:
            out["cpu_baseline"] = cpu_baseline(cfg, a)
        except Exception as e:      # noqa: BLE001
            out["cpu_baseline"] = {"error": f"{type(e).__name__}: {e}"[:300]}
    if rank == 0 and world == 1 and not a.no_side and a.config == "r2r":
        out["side_configs"] = side_configs(a)
    log("done")

    if rank == 0 and a.save_gemm_tuning:
        log(f"gemm tuning table: {ops.save_gemm_tuning_table(a.save_gemm_tuning)} rows -> {a.save_gemm_tuning}")
    if world > 1 or force:
        dist.destroy_process_group()
    faulthandler.cancel_dump_traceback_later()
    if rank == 0:
        # after the teardown, and after whatever C libraries still hold in their stdio buffers (RCCL's version banner):
        # the JSON is the last line on stdout
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:       # noqa: BLE001
            pass
        detail_path = None
        try:
            with open(a.detail, "w") as f:
                json.dump(out, f, indent=1)
            detail_path = a.detail
        except OSError as e:
            log(f"detail record not written ({e!r})")
        sys.stderr.write("[bench detail] " + json.dumps(out) + "\n")
        sys.stderr.flush()
        print(json.dumps(compact_line(out, detail_path)), flush=True)


def side_configs(a):
    """The other single-GPU configurations of BASELINE.json, each in a short child process on the same GPU (the parent
    is idle by now): configs[3]'s per-rank shape (RxR: xlm-roberta vocabulary of 250 002 tokens, 160-token
    instructions, batch 32), the continuous-environment fork's model, and configs[4] (fine-tune rollout, batch 32,
    15 navigation steps: scripts/ft_r2r.bash:37 --max_action_len 15).  Side measurements, not the bench metric; skipped
    one by one once the whole run has used its time budget."""
    import subprocess
    budget_s = float(os.environ.get("BEVBERT_BENCH_SIDE_BUDGET_S", "300"))
    common = ["--steps", "22", "--warmup", "11", "--no-cpu-baseline", "--no-kernel-pass", "--no-stream", "--no-side", "--no-fwd"]
    jobs = [("rxr_b32_len160", [sys.executable, os.path.join(ROOT, "bench.py"), "--config", "rxr", "--txt-len", "160",
                                "--batch", "32"] + common),
            ("ce_b64", [sys.executable, os.path.join(ROOT, "bench.py"), "--config", "ce"] + common),
            # the reference's DEFAULT precision (configs/r2r_pretrain.json "fp16": false): fp32 tensors, fp32 library GEMMs,
            # attention on the fp32 matrix instructions (attn_f32.hip)
            ("r2r_b64_fp32", [sys.executable, os.path.join(ROOT, "bench.py"), "--dtype", "fp32"] + common),
            # the other bf16 mode: every activation rounded to bf16 (the bench line of rounds 1-5) when the line runs the fp32
            # residual stream of torch.autocast (train_r2r.py:256-258), and the other way round
            ("r2r_b64_bf16_residual_" + ("bf16" if a.residual == "fp32" else "fp32"),
             [sys.executable, os.path.join(ROOT, "bench.py"), "--residual", "bf16" if a.residual == "fp32" else "fp32"] + common),
            ("finetune_rollout_b32_15steps_infer", [sys.executable, os.path.join(ROOT, "scripts", "bench_nav.py"), "--batch", "32",
                                                    "--steps", "15", "--iters", "4", "--warmup", "3", "--mode", "infer"]),
            # the same rollout with the agent's action feedback: logits read back and argmaxed on the host every step
            ("finetune_rollout_b32_15steps_infer_feedback", [sys.executable, os.path.join(ROOT, "scripts", "bench_nav.py"),
                                                             "--batch", "32", "--steps", "15", "--iters", "4", "--warmup", "3",
                                                             "--mode", "infer", "--feedback"]),
            # training rollout: 15 forwards, ONE backward through all of them, clip + AdamW (agent.py:339-420): forward and
            # backward graph per step of the episode (nav_static.NavTrainRunner)
            ("finetune_rollout_b32_15steps_train", [sys.executable, os.path.join(ROOT, "scripts", "bench_nav.py"), "--batch", "32",
                                                    "--steps", "15", "--iters", "3", "--warmup", "3", "--mode", "train"])]
    res = {}
    for name, cmd in jobs:
        if time.perf_counter() - T_START > budget_s:
            res[name] = {"skipped": f"time budget of {budget_s:.0f} s for the whole bench run used up"}
            continue
        log(f"side config {name}")
        try:
            p = subprocess.run(cmd, capture_output=True, text=True, timeout=240)
            line = [ln for ln in p.stdout.strip().splitlines() if ln.startswith("{")]
            d = json.loads(line[-1])
            keep = ("value", "unit", "ms_per_step", "dtype", "step_launch", "config", "final_loss", "ms_per_nav_step", "workload",
                    "episodes_per_s", "host_map_bookkeeping_ms_per_nav_step", "ms_per_episode_batch", "map", "action_feedback",
                    "neighbour_bound_overflow")
            res[name] = {k: d[k] for k in keep if k in d}
        except Exception as e:      # noqa: BLE001 -- a side figure must not cost the bench line
            res[name] = {"error": repr(e)[:300]}
    return res


class _CollateStream(torch.utils.data.IterableDataset):
    """Fresh collates for the sustained run, produced in DataLoader worker processes like the reference's task loaders
    (pretrain_src/data/loader.py:122-160 build_dataloader: num_workers = n_workers, collate_fn = the task's collate):
    step i draws ``batch`` samples from a pool of pre-generated synthetic samples (the pool stands in for the dataset
    reads; grid features are rows of the device-resident store), collates them (padding, stacking, fresh MLM masking:
    pretrain_src/data/tasks.py:116-163) and names ``batch`` random rows of the grid-feature store."""

    def __init__(self, cfg, samples, cycle, n_total, batch, seed, n_rows, ship_grid):
        self.cfg, self.samples, self.cycle, self.n_total, self.batch = cfg, samples, cycle, n_total, batch
        self.seed, self.n_rows, self.ship_grid = seed, n_rows, ship_grid

    def __iter__(self):
        import numpy as np
        from vln_bevbert_amd import synthetic
        info = torch.utils.data.get_worker_info()
        w, W = (info.id, info.num_workers) if info is not None else (0, 1)
        torch.set_num_threads(1)
        for i in range(w, self.n_total, W):
            t = self.cycle[i % len(self.cycle)]
            rng = np.random.default_rng(self.seed + i)
            pick = rng.integers(0, len(self.samples), self.batch)
            b = synthetic.collate([self.samples[int(j)] for j in pick], self.cfg, t, rng, sems_as="ids")
            keys = None if self.ship_grid else [f"s_{int(r)}" for r in rng.integers(0, self.n_rows, self.batch)]
            yield t, b, keys


def sustained(cfg, a, trainer, cycle, tasks, dev, resident_ms):
    """Throughput with the loader's work INSIDE the measured loop (the headline figure replays resident batches).

    ``--loader-workers`` DataLoader worker processes collate a FRESH batch for every step (``_CollateStream``; pinned by
    the DataLoader's pin-memory thread); the producer thread of loader.StreamingLoader builds everything the reference
    computes on the host inside its forward (StaticBatch.plan: masked-token positions, SAP fusion table, global-map
    aggregation CSR), picks the shape bucket (loader.BucketManager), refills one of the bucket's two buffer sets on a
    copy stream and hands it to the training thread, which waits for the copy on the compute stream and runs the step
    (captured graph of that buffer set once it exists, eager before).  Grid features come as row numbers of a
    device-resident feature_store.GridFeatureStore (--ship-grid: as 462 MB of fp32 per batch over PCIe, the reference's
    way).  ``loader_ms_per_batch`` is the producer thread's own work per batch (plan + refill), without the time it
    waits for a buffer set to be released (``loader_wait_ms_per_batch``) or for the workers (``collate_*``)."""
    import numpy as np
    from vln_bevbert_amd import ops, synthetic
    from vln_bevbert_amd.feature_store import GridFeatureStore
    from vln_bevbert_amd.loader import BucketManager, StreamingLoader
    store = None
    n_rows = 1024
    if not a.ship_grid:
        g = torch.Generator(device=dev).manual_seed(5)
        P = 12 * cfg.grid_hw * cfg.grid_hw
        rgbs = torch.randn(n_rows, P, 768, device=dev, dtype=torch.float16, generator=g)
        depths = torch.rand(n_rows, 12, cfg.grid_hw, cfg.grid_hw, device=dev, generator=g) * 0.6
        depths = depths * (torch.rand(depths.shape, device=dev, generator=g) > 0.05)
        sems = torch.randint(0, max(1, cfg.sem_classes), (n_rows, P), device=dev, generator=g).to(torch.uint8)
        store = GridFeatureStore([f"s_{i}" for i in range(n_rows)], rgbs, depths, sems, dev)
    # the "dataset": a pool of samples (fixed shapes: T = 5, L = txt_len; --ragged: T in [1,7], text in [L/2, L], 36..38 views)
    rng = np.random.default_rng(11)
    n_pool = 64 if a.ship_grid else 256
    samples = []
    for i in range(n_pool):
        T = int(rng.integers(1, 8)) if a.ragged else 5
        L = int(rng.integers(a.txt_len_min or a.txt_len // 2, a.txt_len + 1)) if a.ragged else a.txt_len
        samples.append(synthetic.make_sample(rng, i, cfg, T, L, ragged_views=a.ragged, grid=a.ship_grid))
    t0 = time.perf_counter()
    for t in tasks:
        synthetic.collate(samples[:a.batch] if len(samples) >= a.batch else (samples * a.batch)[:a.batch], cfg, t, rng, sems_as="ids")
    collate_ms = 1000.0 * (time.perf_counter() - t0) / len(tasks)
    # BEVBERT_OVERLAP_ZERO_SUSTAINED=0: the gradient-arena fill in line at the start of the step (A/B knob).  Until the
    # loader's copy stream was moved off the compute stream's hardware queue (loader.pick_copy_stream) the side-stream fill
    # cost a live-loader run what it hid (18.14 vs 17.92 ms); with it: 18.06 beside the forward vs 18.36 in line (r06z).
    # The buffer sets of this run are captured with the setting in force here.
    overlap_zero_was = trainer.overlap_zero
    trainer.overlap_zero = os.environ.get("BEVBERT_OVERLAP_ZERO_SUSTAINED", "1") == "1" and overlap_zero_was
    mgr = BucketManager(cfg, dev, depth=2, max_buckets=64, grid_store=store)
    # warm-up: every (bucket, buffer set) has to be seen GRAPH_WARMUP + 1 times before its step is a replay
    per_task_uses = {t: max(1, cycle.count(t)) for t in tasks}
    n_buckets_guess = {t: (6 if a.ragged else 1) for t in tasks}
    n_warm = max(len(cycle) * -(-(n_buckets_guess[t] * 2 * (trainer.GRAPH_WARMUP + 1)) // per_task_uses[t]) for t in tasks)
    n_warm = min(n_warm, 40 * len(cycle))
    n_total = n_warm + a.stream_steps
    workers = max(0, a.loader_workers)
    ds = _CollateStream(cfg, samples, cycle, n_total, a.batch, 7000, n_rows, a.ship_grid)
    # pin_memory=True: the DataLoader's pin thread takes the batches out of the workers' shared memory and pins them
    # ahead of the producer thread (measured, round 4, one box: sustained 18.4 ms/step = the resident-batch step with it;
    # 39.5 ms/step when the producer thread itself copies out of the shared-memory segments -- 10 ms per batch of page
    # faults).  Pageable tensors from other sources are staged through pinned buffers the buffer sets own (StaticBatch._staged).
    dl = torch.utils.data.DataLoader(ds, batch_size=None, num_workers=workers,
                                     pin_memory=os.environ.get("BEVBERT_BENCH_DL_PIN", "1") == "1",
                                     prefetch_factor=2 if workers else None, persistent_workers=False)
    loader = StreamingLoader(iter(dl), mgr, prefetch=1)
    plans0 = ops.gemm_plan_count() if hasattr(ops, "gemm_plan_count") else None
    it = iter(loader)
    replayed = eager = 0
    t0 = None
    stats0 = None
    marks = [] if os.environ.get("BEVBERT_STEP_EVENTS") == "1" else None     # diagnosis: device time of every step + gaps
    host_log, gc_log = [], []
    if marks is not None:
        import gc
        mgr.trace = []

        def _gc_cb(phase, info, _t=[0.0]):
            if phase == "start":
                _t[0] = time.perf_counter()
            else:
                gc_log.append((_t[0], time.perf_counter() - _t[0], info.get("generation")))
        gc.callbacks.append(_gc_cb)
    for i in range(n_total):
        if i == n_warm:
            torch.cuda.synchronize()
            if os.environ.get("BEVBERT_QUIET_GC", "1") == "1":
                from vln_bevbert_amd.loader import quiet_gc
                quiet_gc()                       # an 85 ms generation-2 pass in the timed region costs 1 - 6 % of 33 steps
            stats0 = dict(mgr.stats)
            replayed = eager = 0
            t0 = time.perf_counter()
        t_a = time.perf_counter()
        task, sb = next(it)
        t_b = time.perf_counter()
        was_graph = sb.graph is not None
        if marks is not None and i >= n_warm:
            marks.append((task, torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)))
            marks[-1][1].record()
        trainer.step(task, sb)
        if marks is not None and i >= n_warm:
            marks[-1][2].record()
            host_log.append((t_a, t_b, time.perf_counter(), task))
        loader.release(sb)
        replayed += was_graph
        eager += not was_graph
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    loader.close()
    del it, dl
    st = {k: mgr.stats[k] - stats0.get(k, 0) for k in mgr.stats}
    ms = 1000.0 * dt / a.stream_steps
    n = max(1, a.stream_steps)
    res = {"samples_per_s": round(a.stream_steps * a.batch / dt, 2), "ms_per_step": round(ms, 3),
           "vs_resident": round(resident_ms / ms, 4), "steps": a.stream_steps, "warmup_steps": n_warm,
           "batches": (f"ragged (T in [1,7] panoramas, text in [{a.txt_len_min or a.txt_len // 2}, {a.txt_len}] tokens padded to the batch "
                       "maximum rounded up to 16, 36..38 views)") if a.ragged else "fixed shapes (T = 5, L = %d)" % a.txt_len,
           "grid_features": "462 MB fp32 per batch over PCIe" if a.ship_grid else f"rows of a {store.nbytes() / 2**30:.1f} GiB device-resident store",
           "source": f"a fresh collate per step ({n_pool}-sample pool, {workers} DataLoader worker processes + its pin-memory "
                     f"thread); host-side index building and refill every step",
           "collate_ms_per_batch_one_process": round(collate_ms, 2), "collate_workers": workers,
           "buckets": len(mgr.buckets), "buckets_created_in_timed_region": st.get("buckets_created", 0),
           "captured_graphs": mgr.captured_graphs(), "steps_replayed": replayed, "steps_eager": eager,
           "loader_ms_per_batch": round(1000.0 * (st.get("loader_s", 0.0) - st.get("wait_s", 0.0)) / n, 3),
           "loader_wait_ms_per_batch": round(1000.0 * st.get("wait_s", 0.0) / n, 3),
           "producer_waits_for_collate_ms_per_batch": round(1000.0 * st.get("source_s", 0.0) / n, 3),
           "producer_waits_for_consumer_ms_per_batch": round(1000.0 * st.get("queue_s", 0.0) / n, 3),
           "h2d_MB_per_step": round(st.get("bytes_h2d", 0) / n / 1e6, 2)}
    if plans0 is not None:
        res["gemm_plans_added"] = ops.gemm_plan_count() - plans0
    res["arena_fill"] = "beside the forward (side stream)" if trainer.overlap_zero else "in line"
    res["copy_stream_probe"] = mgr.copy_stream_probe
    if marks:
        import gc
        gc.callbacks[:] = [c for c in gc.callbacks if getattr(c, "__name__", "") != "_gc_cb"]
        res["step_events"] = _step_event_digest(marks)
        gaps = [marks[j - 1][2].elapsed_time(marks[j][1]) for j in range(1, len(marks))]
        j = max(range(len(gaps)), key=gaps.__getitem__) + 1            # the step that started late
        if gaps[j - 1] > 2.0:
            ms = lambda t: round(1000.0 * (t - t0), 2)
            lo_t, hi_t = host_log[max(0, j - 4)][0], host_log[min(len(host_log) - 1, j + 1)][2]
            res["step_events"]["worst_gap"] = {
                "ms": round(gaps[j - 1], 2), "before_step": j,
                "consumer_next_wait_step_ms": [(k, hl[3], ms(hl[0]), round(1000 * (hl[1] - hl[0]), 2), round(1000 * (hl[2] - hl[1]), 2))
                                               for k, hl in enumerate(host_log) if max(0, j - 4) <= k <= j + 1],
                "producer_enter_wait_total_ms": [(ms(tr[0]), round(1000 * (tr[1] - (mgr.trace[q - 1][1] if q else 0.0)), 2),
                                                  round(1000 * (tr[2] - tr[0]), 2), tr[3])
                                                 for q, tr in enumerate(mgr.trace) if lo_t - 0.05 <= tr[0] <= hi_t],
                "gc_start_ms_dur_ms_gen": [(ms(g[0]), round(1000 * g[1], 2), g[2]) for g in gc_log if lo_t - 0.05 <= g[0] <= hi_t and g[1] > 0.002]}
    trainer.overlap_zero = overlap_zero_was
    del store
    return res


def _step_event_digest(marks):
    """marks: (task, start event, end event) per step, recorded on the compute stream around trainer.step -> device time of
    the steps by task and the idle time between the end of one step and the start of the next."""
    by, gaps = {}, []
    for j, (t, e0, e1) in enumerate(marks):
        by.setdefault(t, []).append(e0.elapsed_time(e1))
        if j:
            gaps.append(marks[j - 1][2].elapsed_time(e0))
    return {"device_ms_by_task": {t: round(sum(v) / len(v), 3) for t, v in by.items()},
            "device_ms_per_step": round(sum(sum(v) for v in by.values()) / len(marks), 3),
            "gap_ms_per_step": round(sum(gaps) / max(1, len(gaps)), 3), "gap_ms_max": round(max(gaps or [0.0]), 3)}


def cpu_baseline(cfg, a):
    """The CPU oracle (oracle/bevbert_ref.py, a restatement pinned to the reference by golden vectors) on the host cores.

    ``forward``: BASELINE.md section 3's protocol -- forward only, SAP and MLM, batch 2 (BASELINE.json configs[0]) and
    16, one thread and all usable cores, fp32, eval; 2 warm-ups + the median of up to 10 repetitions, cut short by a time
    budget per cell (the slow cells say how many repetitions they got).  ``value`` (the contract field) stays the
    training figure: fwd + bwd + AdamW at batch 8 on all cores, the same work the GPU line measures."""
    from oracle import bevbert_ref as R
    from vln_bevbert_amd import synthetic, weights
    from vln_bevbert_amd.pretrain_cmt import GlocalTextPathCMTPreTraining
    n = a.cpu_threads or usable_cores()
    torch.set_num_threads(n)
    shapes = {k: tuple(v.shape) for k, v in GlocalTextPathCMTPreTraining(cfg).state_dict().items()}
    sd = {k: v.requires_grad_(True) for k, v in weights.fill_state_dict(shapes).items()}
    params = list({id(v): v for v in sd.values()}.values())
    m = [torch.zeros_like(p) for p in params]
    v = [torch.zeros_like(p) for p in params]
    B = 8
    t_total, n_samples, budget_s = 0.0, 0, 20.0
    order = ("sap", "mlm", "sap", "mlm", "masksem", "mlm", "sap", "mlm", "sap")      # first one is the untimed warm-up
    order = tuple(t if t in cfg.pretrain_tasks else "mlm" for t in order)
    for it, task in enumerate(order):
        if t_total > budget_s:
            break
        b = synthetic.make_batch(cfg, task, B, seed=4000 + it, txt_len=a.txt_len)
        t0 = time.perf_counter()
        loss = R.pretrain_forward(sd, cfg, b, task).mean()
        grads = torch.autograd.grad(loss, params, allow_unused=True)
        with torch.no_grad():
            for p, g, mm, vv in zip(params, grads, m, v):
                if g is not None:
                    R.adamw_step(p, g, mm, vv, it + 1, 5e-5, 0.01)
        dt = time.perf_counter() - t0
        log(f"  cpu step {it} ({task}): {dt:.2f} s")
        if it > 0:                         # first iteration pages everything in
            t_total += dt
            n_samples += B
    out = {"value": round(n_samples / t_total, 3), "unit": "samples/s", "cores": n, "kind": "port",
           "sample": f"{n_samples // B} steps ({', '.join(order[1:1 + n_samples // B])}) of batch {B}, same shapes, "
                     f"fp32, dropout off, fwd+bwd+AdamW, {t_total:.1f} s of torch CPU work with {n} threads after "
                     "1 untimed warm-up step"}
    # ---- forward only, BASELINE.md section 3
    gflop = {"sap": 50.81, "mlm": 32.25}          # per sample, BASELINE.md section 2 (FlopCounterMode on the reference)
    sdf = {k: t.detach() for k, t in sd.items()}
    cells, cell_budget = [], 5.0
    for task in ("sap", "mlm"):
        if task not in cfg.pretrain_tasks:
            continue
        for bsz in (2, 16):
            batch = synthetic.make_batch(cfg, task, bsz, seed=4100 + bsz, txt_len=a.txt_len)
            for threads in sorted({1, n}):
                torch.set_num_threads(threads)
                times, warm, spent = [], 0, 0.0
                with torch.no_grad():
                    while len(times) < 10 and (spent < cell_budget or not times):
                        t0 = time.perf_counter()
                        R.pretrain_forward(sdf, cfg, batch, task)
                        dt = time.perf_counter() - t0
                        spent += dt
                        if warm < 2 and spent + 2 * dt < cell_budget:      # warm-ups only while the cell can afford them
                            warm += 1
                        else:
                            times.append(dt)
                times.sort()
                med = times[len(times) // 2]
                cells.append({"task": task, "batch": bsz, "threads": threads, "warmups": warm, "reps": len(times),
                              "median_s": round(med, 4), "min_s": round(times[0], 4),
                              "samples_per_s": round(bsz / med, 3), "gflops": round(gflop[task] * bsz / med, 1)})
                log(f"  cpu forward {task} B={bsz} threads={threads}: median {med:.3f} s over {len(times)} reps")
    torch.set_num_threads(n)
    try:
        with open("/proc/cpuinfo") as f:
            model_name = next((ln.split(":", 1)[1].strip() for ln in f if ln.startswith("model name")), "?")
    except Exception:       # noqa: BLE001
        model_name = "?"
    out["forward"] = {"protocol": "BASELINE.md section 3: forward only, fp32, eval, 2 warm-ups + median of up to 10 "
                                  f"repetitions, {cell_budget:.0f} s budget per cell", "host_cpu": model_name,
                      "usable_cores": n, "cells": cells}
    out["train"] = {"value": out["value"], "unit": "samples/s", "cores": n, "what": "fwd + bwd + AdamW, batch 8"}
    return out


if __name__ == "__main__":
    main()
