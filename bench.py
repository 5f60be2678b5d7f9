#!/usr/bin/env python3
"""bench.py -- throughput of the fused mixture-of-tokenizers embedding front-end on MI355X.

One "step" = one pass of the hot path over one synthetic batch: tokens (int32, resident in HBM)
-> token->byte table gather -> pull-from-left -> token/byte embedding gathers -> sum -> rms-norm
-> x (fp32), i.e. ONE launch of mot_embed_mix_fwd (include/mot.h).  Default workload is
BASELINE.json configs[3]: B x T = 256 x 2048, GPT-2 vocab 50257, bpt 16, d_model 768 (byte dim
48), FineWeb-shaped ids.

Multi-GPU (`--gpus N`): one process per GPU.  Started under torchrun (RANK / WORLD_SIZE in the
environment) the process is one rank; started plainly (`python3 bench.py --gpus N`) the parent
launches the N ranks itself as child processes BEFORE making any GPU call, and relays rank 0's
JSON line.  Default for N > 1 is STRONG scaling -- config 4 as BASELINE words it: ONE 256 x 2048
batch whose rows are split over the ranks exactly as the reference's loader slices them
(train_gpt.py:795-805) -- `--scaling weak` runs the full batch on every rank instead.  The path
has no data-path collective; the only RCCL traffic is the all-reduce of the int64[4] byte
statistics (and of the timings) after the timed region.

Prints ONE JSON line (rank 0).  `roofline.achieved` = algorithmic bytes per launch / average
launch duration measured with HIP events on the launch stream over the timed region (per rank;
the slowest rank's duration is used); `cpu_baseline` = the CPU oracle (a C/OpenMP port of the
reference path) timed on this host on a bounded sample; `extra` (N = 1) = the same kernel where
caches cannot help: uniform ids, a token table larger than the Infinity Cache, the 65 536-token
shard each GPU sees at 8-way strong scaling.
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time
from pathlib import Path

np = torch = None   # imported by _imports(), AFTER main() has decided whether this process only spawns the ranks


def _imports():
    """numpy + torch, bound as module globals.  The spawning parent of `--gpus N` never gets here: it imports neither."""
    global np, torch
    import numpy as _np
    import torch as _torch
    np, torch = _np, _torch


REPO = Path(__file__).resolve().parent
sys.path.insert(0, str(REPO))
sys.path.insert(0, str(REPO / "tests"))

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); measured copy ceiling ~6300

WORKLOADS = {
    # name: (B, T, vocab, bpt, d_model, byte_dim, byte_vocab, mode)
    "c4": (256, 2048, 50257, 16, 768, 48, 458, "sum"),     # headline: BASELINE configs[3]
    "c2": (64, 1024, 50257, 16, 768, 48, 458, "sum"),      # configs[1]
    "c3": (128, 2048, 50257, 1, 768, 768, 100277, "dual"),  # configs[2]: GPT-2 + cl100k dual-BPE mix: a second TOKEN-level table, one id per position
    "c4big": (256, 2048, 128256, 16, 768, 48, 458, "sum"),  # the headline kernel over a 394 MB token table (> Infinity Cache)
    "c5": (256, 8192, 128256, 8, 2048, 2048, 132, "mean"),  # configs[4]: Llama-3 vocab, d 2048, 8 char slots, full batch
    "c5q": (64, 8192, 128256, 8, 2048, 2048, 132, "mean"),  # a quarter of it (round-1 record)
    # concat + linear mixin (MFMA-bound): (.., d_model, byte_dim, .., mode, token_dim)
    "c2l": (64, 1024, 50257, 16, 768, 32, 458, "concat_linear", 256),     # configs[1] CONCAT dims (K = 768)
    "prodl": (64, 1024, 50257, 16, 1024, 48, 458, "concat_linear", 256),  # experiments100_000steps.sh dims (K = 1024)
}
F32_MFMA_PEAK_TFLOPS = 157.3  # dense fp32 matrix peak, MI355X_MICROARCH.md
# sources whose change invalidates the committed PMC traffic figure of the fused SUM kernel
TRAFFIC_SOURCES = ("mixture-of-tokenizers_amd/csrc/mot_embed.hip", "mixture-of-tokenizers_amd/csrc/mot_wave.hpp",
                   "mixture-of-tokenizers_amd/csrc/mot_mix.hpp", "mixture-of-tokenizers_amd/csrc/mot_tile.hpp")   # what the fused kernel is compiled from


def source_sha16() -> str:
    h = hashlib.sha256()
    for rel in TRAFFIC_SOURCES:
        h.update((REPO / rel).read_bytes())
    return h.hexdigest()[:16]


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="c4", choices=sorted(WORKLOADS))
    ap.add_argument("--ids", default="fused", choices=["fused", "given"],
                    help="fused: byte ids produced inside the kernel; given: int64 ids precomputed (module-level path)")
    ap.add_argument("--uniform-ids", action="store_true", help="uniform token ids (no-reuse worst case)")
    ap.add_argument("--scaling", default=None, choices=["weak", "strong"],
                    help="strong (default for --gpus > 1): the workload's B rows are split over the ranks (config 4 as worded: one "
                         "256x2048 batch sharded over the GPUs, train_gpt.py:795-805); weak: every rank runs the full batch")
    ap.add_argument("--dtype", default="f32", choices=["f32", "bf16"], help="table/output element type (compute is fp32)")
    ap.add_argument("--backward", action="store_true", help="also time the backward launch (sum workloads) and report it")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the secondary records (uniform ids, table > Infinity Cache, 65 536-token shard)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="process-group backend of the N-rank run: nccl (= RCCL over xGMI, the real thing) or gloo (collectives on CPU "
                         "copies of the few counters: rehearsals on a box with fewer GPUs than ranks)")
    ap.add_argument("--one-device", action="store_true",
                    help="REHEARSAL: every rank uses cuda:0 (needs --backend gloo; RCCL refuses two ranks on one GPU). Exercises rank > 0 "
                         "of the row slicing / timing / counter aggregation on a one-GPU box; the line is marked \"rehearsal\": true")
    ap.add_argument("--dry-run", action="store_true",
                    help="launcher / rendezvous / aggregation only: gloo on the CPU, no GPU call, no kernel (CPU tests of the N-rank path)")
    args = ap.parse_args(argv)
    if args.scaling is None:
        args.scaling = "strong" if args.gpus > 1 else "weak"
    return args


# ------------------------------------------------------------------------------------------------
# self-launch: `python3 bench.py --gpus N` without torchrun
# ------------------------------------------------------------------------------------------------
def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def visible_gpus() -> int | None:
    """GPUs this process may use, counted WITHOUT touching the HIP runtime: the KFD topology in sysfs (nodes with SIMDs are
    GPUs; CPU nodes report simd_count 0), cut down by HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES lists when set.  None when sysfs
    is unreadable (then the pre-check is skipped and a rank that finds no device fails by itself)."""
    try:
        n = 0
        for prop in Path("/sys/class/kfd/kfd/topology/nodes").glob("*/properties"):
            for ln in prop.read_text().splitlines():
                k, _, v = ln.partition(" ")
                if k == "simd_count" and int(v) > 0:
                    n += 1
        if n == 0:
            return None
    except Exception:
        return None
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        lst = os.environ.get(var)
        if lst is not None:
            n = min(n, len([x for x in lst.split(",") if x.strip() != ""]))
    return n


def spawn_ranks(args, argv) -> int:
    """Parent of a plain `--gpus N` run: starts the N ranks as child processes (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in
    their environment, as torchrun would) and relays rank 0's JSON line.  This process makes NO torch.cuda / HIP call at all
    (a runtime initialised here would be inherited across fork + exec by every rank): the device pre-check reads sysfs."""
    assert "torch" not in sys.modules, "the spawning parent must not import torch (no HIP runtime before the ranks start)"
    if not args.dry_run and not args.one_device:
        have = visible_gpus()
        if have is not None and have < args.gpus:
            print(f"bench.py: --gpus {args.gpus} but only {have} visible", file=sys.stderr)
            return 2
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), WORLD_SIZE=str(args.gpus),
               HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    procs = []
    for r in range(args.gpus):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, str(Path(__file__).resolve())] + list(argv), env=e,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    out0, _ = procs[0].communicate()
    rcs = [p.wait() for p in procs]
    for ln in (out0 or "").splitlines():     # the contract is ONE JSON line on stdout: library chatter of rank 0 goes to stderr
        (sys.stdout if ln.lstrip().startswith("{") else sys.stderr).write(ln + "\n")
    sys.stdout.flush()
    bad = [(r, c) for r, c in enumerate(rcs) if c]
    if bad:
        print(f"bench.py: ranks failed (rank, exit code): {bad}", file=sys.stderr)
        return 1
    return 0


def make_inputs(wl, device, seed, uniform, rows=None, row0=0, vocab_override=None):
    """Synthetic inputs of one rank: `rows` rows starting at global row `row0` of the seeded batch (strong scaling slices one
    global batch, as the loader's rank slice does)."""
    import golden_inputs as gi
    B, T, vocab, bpt, D, Db, Vb, mode = WORKLOADS[wl][:8]
    vocab = vocab_override or vocab
    rows = rows or B
    # the ids are always drawn for the WHOLE global batch and then sliced: the generators draw several arrays one after the other,
    # so a draw of fewer rows would give different values for the same rows (sharded must equal unsharded, row for row)
    full = max(B, row0 + rows)
    Dt = WORKLOADS[wl][8] if mode == "concat_linear" else D
    g = torch.Generator(device=device).manual_seed(12345)                       # tables are replicated: same on every rank
    tok_table = torch.randn((vocab, Dt), generator=g, device=device, dtype=torch.float32)
    byte_table = torch.randn((Vb, Db), generator=g, device=device, dtype=torch.float32)
    if mode in ("sum", "concat_linear"):
        try:
            tab = gi.widen_left_pad(gi.load_real_ttb8(), bpt)     # real GPT-2 token->char table (data fixture)
            ttb_kind = "gpt2 ttb_8_left_pad widened to 16"
            if vocab > tab.shape[0]:                               # larger synthetic vocabulary: the real rows, repeated
                tab = np.concatenate([tab] * (-(-vocab // tab.shape[0])))[:vocab]
                ttb_kind += f", tiled to {vocab} rows"
        except FileNotFoundError:
            tab = gi.synth_ttb(5, vocab, bpt, "left")
            ttb_kind = "synthetic"
        toks = gi.fineweb_like_tokens(seed, full, T, vocab=vocab, uniform=uniform)[row0:row0 + rows]
        weight = None
        if mode == "concat_linear":
            K = Dt + bpt * Db
            bound = (3 ** 0.5) * 0.5 * K ** -0.5       # CastedLinear init, train_gpt.py:179-183
            weight = (torch.rand((D, K), generator=g, device=device, dtype=torch.float32) * 2 - 1) * bound
        return dict(toks=toks, tab=tab, tok_table=tok_table, byte_table=byte_table, ttb_kind=ttb_kind, weight=weight)
    if mode == "dual":   # no counterpart in the reference: synthetic ids, both vocabularies FineWeb-shaped
        toks = gi.fineweb_like_tokens(seed, full, T, vocab=vocab, uniform=uniform)[row0:row0 + rows]
        ids2 = gi.fineweb_like_tokens(seed + 7, full, T, vocab=Vb, uniform=uniform)[row0:row0 + rows].astype(np.int64)
        return dict(toks=toks, chars=ids2, tok_table=tok_table, byte_table=byte_table, ttb_kind="n/a (second id tensor given)")
    rs = np.random.RandomState(seed)
    toks = rs.randint(0, vocab, size=(full, T)).astype(np.int32)[row0:row0 + rows]
    chars = rs.randint(0, Vb, size=(full, T * bpt)).astype(np.int64)[row0:row0 + rows]
    return dict(toks=toks, chars=chars, tok_table=tok_table, byte_table=byte_table, ttb_kind="n/a")


def algorithmic_bytes_per_token(wl, ids_mode, e=4):
    """fully fused R = 4 + 2*bpt + e*Dt, W = e*Dm; module-level path reads int64 ids."""
    B, T, vocab, bpt, D, Db, Vb, mode = WORKLOADS[wl][:8]
    if mode == "concat_linear":
        return 4 + 2 * bpt + e * WORKLOADS[wl][8] + e * D
    if mode == "sum":
        r = 4 + (2 * bpt if ids_mode == "fused" else 8 * bpt) + e * D
    elif mode == "dual":
        r = 4 + 8 + 2 * e * D          # two table rows per token
    else:
        r = 4 + 8 * bpt + e * D
    return r + e * D


def shard_reference(wl, args, tokens):
    """What ONE GPU reaches on a launch of this many tokens (profiles/shard_reference.json: single-GPU runs of the shard sizes of
    2/4/8-way strong scaling, recorded with tools/bench_shard.py): every rank of a strong-scaling run launches exactly this, so
    `roofline.frac` of an N-GPU line should sit at this figure (also in `extra.shard_65536_tokens` of the N = 1 line)."""
    if wl != "c4" or args.ids != "fused" or args.dtype != "f32" or args.uniform_ids:
        return None
    try:
        ref = json.loads((REPO / "profiles" / "shard_reference.json").read_text())
        r = ref["tokens"].get(str(tokens))
        return None if r is None else dict(r, source=ref["source"])
    except Exception:
        return None


def usable_cores() -> int:
    """Threads the CPU baseline may really use: affinity, capped by the cgroup CPU quota and by the
    GPU box's per-GPU CPU share (16), so OpenMP does not oversubscribe."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = Path("/sys/fs/cgroup/cpu.max").read_text().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, min(n, int(os.environ.get("MOT_CPU_THREADS", "16"))))


def cpu_baseline(wl, inp, seconds):
    """The oracle (oracle/mot_oracle.c, OpenMP) on this host: tokens_to_bytes + pull_from_left +
    gather/sum/rms-norm in fp32 on rows of the same workload, repeated for ~`seconds`."""
    from oracle import oracle as orc
    import golden_inputs as gi
    B, T, vocab, bpt, D, Db, Vb, mode = WORKLOADS[wl][:8]
    cores = usable_cores()
    orc.set_threads(cores)
    Et, Eb = inp["tok_table"].float().cpu().numpy(), inp["byte_table"].float().cpu().numpy()
    rows = min(len(inp["toks"]), 32 if mode != "mean" else 4)
    toks = inp["toks"][:rows]

    def once():
        if mode == "sum":
            padded = orc.tokens_to_bytes(toks, inp["tab"].astype(np.float32))
            pulled = orc.pull_from_left(padded, bpt, gi.PAD, gi.EOT)
            orc.embed_mix(toks, pulled, None, Et, Eb, mode="sum", bpt=bpt, dtype=np.float32, norm_out=True)
        elif mode == "concat_linear":
            padded = orc.tokens_to_bytes(toks, inp["tab"].astype(np.float32))
            pulled = orc.pull_from_left(padded, bpt, gi.PAD, gi.EOT)
            orc.embed_mix(toks, pulled, None, Et, Eb, mode="concat_linear", bpt=bpt, weight=inp["weight"].float().cpu().numpy(),
                          dtype=np.float32, norm_tok=True, norm_byte=True, norm_out=True)
        elif mode == "dual":
            orc.embed_mix(toks, inp["chars"][:rows], None, Et, Eb, mode="sum", bpt=1, dtype=np.float32, norm_out=True)
        else:
            orc.embed_mix(toks, inp["chars"][:rows], None, Et, Eb, mode="mean", bpt=bpt, dtype=np.float32)

    once()
    t0 = time.perf_counter()
    reps = 0
    while True:
        once()
        reps += 1
        el = time.perf_counter() - t0
        if el >= seconds or reps >= 5000:
            break
    return dict(value=rows * T * reps / el, unit="tokens/s", cores=cores, kind="port",
                sample=f"{reps} passes over {rows}x{T} tokens of the same workload in {el:.1f} s "
                       f"(oracle/mot_oracle.c, fp32, OpenMP {cores} threads); calibration in the build container "
                       f"(oracle/calibrate_cpu.py, 8 threads): this port runs 8.3x FASTER than the reference's own eager-PyTorch CPU path")


def timed_launches(step, n, warm=5):
    """Average duration (ms) of n back-to-back launches of `step`, HIP events on the current (launch) stream."""
    for _ in range(warm):
        step()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        step()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def extra_records(mot, device, dtype, steps, warmup):
    """The headline kernel where caches cannot carry it (N = 1 only, outside the timed region): each record is the same
    fused SUM launch with its own algorithmic bytes / HIP-event time."""
    recs = {}
    cases = (("c4_uniform_ids", "c4", True, None, "no token id repeats inside L2's reach: every row fetch goes to Infinity Cache / HBM"),
             ("c4_table_394MB", "c4big", False, None, "token table 128 256 x 768 fp32 = 394 MB > the 256 MiB Infinity Cache"),
             ("c4_table_394MB_uniform_ids", "c4big", True, None, "the same table, uniform ids: neither L2 nor Infinity Cache can hold the rows"),
             ("shard_65536_tokens", "c4", False, 32, "32 x 2048 tokens: what each GPU runs at 8-way strong scaling of config 4"))
    for name, wl, uniform, rows, why in cases:
        B, T, vocab, bpt, D, Db, Vb, mode = WORKLOADS[wl][:8]
        inp = make_inputs(wl, device, seed=12345, uniform=uniform, rows=rows)
        rows = rows or B
        tdt = torch.float32 if dtype == "f32" else torch.bfloat16
        tt, bt = inp["tok_table"].to(tdt), inp["byte_table"].to(tdt)
        toks, tab = torch.from_numpy(inp["toks"]).to(device), torch.from_numpy(inp["tab"]).to(device)
        out = torch.empty((rows, T, D), dtype=tdt, device=device)
        plan = mot.embed_mix_plan(toks, tt, bt, mode="sum", bpt=bpt, ttb=tab, pull="left", norm_out=True, out=out)
        ms = timed_launches(plan, steps, warm=warmup)      # the headline's own warm-up and step counts
        nbytes = algorithmic_bytes_per_token(wl, "fused", 4 if dtype == "f32" else 2) * rows * T
        gbs = nbytes / (ms * 1e-3) / 1e9
        recs[name] = {"kernel_ms": ms, "tokens_per_s": rows * T / (ms * 1e-3), "achieved_GBps": gbs, "frac": gbs / HBM_PEAK_GBS,
                      "tokens_per_launch": rows * T, "why": why}
        del inp, tt, bt, out, plan
        torch.cuda.empty_cache()
    if dtype == "f32":   # the production dtype beside the fp32 headline: forward and backward of the same batch with bf16 tables and output
        from mixture_of_tokenizers_amd import data_creation as dc
        B, T, vocab, bpt, D, Db, Vb, mode = WORKLOADS["c4"][:8]
        inp = make_inputs("c4", device, seed=12345, uniform=False)
        tt, bt = inp["tok_table"].bfloat16(), inp["byte_table"].bfloat16()
        toks, tab = torch.from_numpy(inp["toks"]).to(device), torch.from_numpy(inp["tab"]).to(device)
        out = torch.empty((B, T, D), dtype=torch.bfloat16, device=device)
        plan = mot.embed_mix_plan(toks, tt, bt, mode="sum", bpt=bpt, ttb=tab, pull="left", norm_out=True, out=out)
        ms = timed_launches(plan, steps, warm=warmup)
        gbs = algorithmic_bytes_per_token("c4", "fused", 2) * B * T / (ms * 1e-3) / 1e9
        F = mot.functional
        ids = dc.pull_from_left(dc.tokens_to_bytes(toks, tab), bpt, 456, 457)
        into = {"tok_table": torch.zeros_like(tt, dtype=torch.float32), "byte_table": torch.zeros_like(bt, dtype=torch.float32)}
        order = F.token_order(toks, tt.shape[0])
        gout = torch.randn_like(out)
        bms = timed_launches(lambda: F.embed_mix_backward(gout, toks, tt, bt, token_order=order, mode="sum", bpt=bpt, ids_a=ids, norm_out=True, into=into),
                             max(1, steps // 4), warm=3)
        recs["c4_bf16_tables"] = {"kernel_ms": ms, "tokens_per_s": B * T / (ms * 1e-3), "achieved_GBps": gbs, "frac": gbs / HBM_PEAK_GBS,
                                  "tokens_per_launch": B * T, "backward_kernel_ms": bms,
                                  "why": "bf16 tables and output (train_gpt.py:1124-1126): algorithmic bytes with 2-byte elements; backward with the token order given"}
        del inp, tt, bt, out, plan, into, gout
        torch.cuda.empty_cache()
    return recs


def dry_run(args, world, rank):
    """N-rank launcher / rendezvous / aggregation without a GPU: gloo on the CPU, the same barrier + max-over-ranks +
    counter all-reduce sequence as the real run, a synthetic step."""
    import torch.distributed as dist
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    B, T = WORKLOADS[args.workload][:2]
    if args.scaling == "strong":
        assert B % world == 0, "batch_size % world_size == 0 (train_gpt.py:795)"
        B //= world
    t0 = time.perf_counter()
    for _ in range(args.steps):
        pass
    elapsed = max(time.perf_counter() - t0, 1e-9)
    counters = torch.tensor([B * T, B * T * WORKLOADS[args.workload][3], 0, 0], dtype=torch.int64)
    if world > 1:
        dist.barrier()
        tmax = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(counters, op=dist.ReduceOp.SUM)
        elapsed = float(tmax.item())
    if rank == 0:
        print(json.dumps({"metric": "mixed-embed tokens/sec at BxT=256x2048; achieved HBM GB/s vs peak", "value": None, "unit": "tokens/s",
                          "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "scaling": args.scaling, "dry_run": True,
                          "config": {"workload": f"{args.workload}: BxT={B}x{T} per rank", "global_tokens_per_step": int(counters[0])},
                          "byte_stats": {"tokens": int(counters[0]), "slots": int(counters[1])}}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args, argv))       # parent: torch is not even imported in this process
    _imports()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    if args.dry_run:
        return dry_run(args, world, rank)
    # MOT_FORCE_DIST=1 exercises the multi-rank code path (RCCL init, barrier, all-reduce) with a single rank
    use_dist = world > 1 or os.environ.get("MOT_FORCE_DIST") == "1"
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
        if args.one_device:
            assert args.backend == "gloo", "--one-device needs --backend gloo (RCCL refuses two ranks on one GPU)"
            local_rank = 0
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo")
    device = torch.device("cuda", local_rank)
    # collectives of the reporting path: on the device over RCCL, or on CPU copies over gloo (rehearsal)
    cdev = device if args.backend == "nccl" else torch.device("cpu")
    torch.cuda.set_device(device)

    import mixture_of_tokenizers_amd as mot
    wl = args.workload
    B, T, vocab, bpt, D, Db, Vb, mode = WORKLOADS[wl][:8]
    row0 = 0
    if args.scaling == "strong":
        assert B % world == 0, "batch_size % world_size == 0 (train_gpt.py:795)"
        B //= world
        row0 = rank * B                                    # rank r owns rows [r*B/W, (r+1)*B/W) of the one global batch
    # loader default seed (train_gpt.py:661); weak scaling: every rank draws its own full batch
    inp = make_inputs(wl, device, seed=12345 + (rank if args.scaling == "weak" else 0), uniform=args.uniform_ids, rows=B, row0=row0)
    toks = torch.from_numpy(np.ascontiguousarray(inp["toks"])).to(device)
    esize = 4
    if args.dtype == "bf16":
        esize = 2
        for k in ("tok_table", "byte_table", "weight"):
            if inp.get(k) is not None:
                inp[k] = inp[k].to(torch.bfloat16)
    out = torch.empty((B, T, D), dtype=inp["tok_table"].dtype, device=device)
    counters = torch.zeros(4, dtype=torch.int64, device=device)
    if mode == "sum":
        tab = torch.from_numpy(inp["tab"]).to(device)
        if args.ids == "fused":
            kwf = dict(mode="sum", bpt=bpt, ttb=tab, pull="left", norm_out=True, out=out)
        else:
            from mixture_of_tokenizers_amd import data_creation as dc
            ids = dc.pull_from_left(dc.tokens_to_bytes(toks, tab), bpt, 456, 457)
            kwf = dict(mode="sum", bpt=bpt, ids_a=ids, norm_out=True, out=out)
    elif mode == "concat_linear":
        tab = torch.from_numpy(inp["tab"]).to(device)
        if args.ids == "fused":
            kwf = dict(mode="concat_linear", bpt=bpt, ttb=tab, pull="left", weight=inp["weight"], norm_tok=True, norm_byte=True,
                       norm_out=True, out=out)
        else:   # the module seam: byte ids precomputed as the reference's loader emits them
            from mixture_of_tokenizers_amd import data_creation as dc
            ids = dc.pull_from_left(dc.tokens_to_bytes(toks, tab), bpt, 456, 457)
            kwf = dict(mode="concat_linear", bpt=bpt, ids_a=ids, weight=inp["weight"], norm_tok=True, norm_byte=True, norm_out=True, out=out)
    elif mode == "dual":
        ids2 = torch.from_numpy(np.ascontiguousarray(inp["chars"])).to(device)
        kwf = dict(mode="sum", bpt=1, ids_a=ids2, norm_out=True, out=out)
    else:
        chars = torch.from_numpy(np.ascontiguousarray(inp["chars"])).to(device)
        lt, lc = torch.tensor(1.0, device=device), torch.tensor(0.5, device=device)
        kwf = dict(mode="mean", bpt=bpt, ids_a=chars, scale_tok=lt, scale_byte=lc, out=out)

    # one bound descriptor per variant: a step is exactly one mot_embed_mix_fwd call on the current stream
    plan = mot.embed_mix_plan(toks, inp["tok_table"], inp["byte_table"], **kwf)
    plan_cnt = None if mode == "mean" else mot.embed_mix_plan(toks, inp["tok_table"], inp["byte_table"], counters=counters, **kwf)

    def step():
        plan()

    def barrier():
        if use_dist:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(args.steps):
        step()
    ev1.record()
    barrier()
    elapsed = time.perf_counter() - t0
    kernel_ms = ev0.elapsed_time(ev1) / args.steps          # HIP events on the launch stream
    # per-launch distribution (median and p10/p90), outside the timed region: 64 individually evented launches
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(64)]
    for a, b in evs:
        a.record(); step(); b.record()
    torch.cuda.synchronize()
    lat = sorted(a.elapsed_time(b) * 1e3 for a, b in evs)
    launch_us = {"p10": lat[6], "p50": lat[32], "p90": lat[57]}
    # what a plain device copy of the same volume reaches on this GPU ("also report against a measured
    # device-copy bandwidth"): read + write of out-sized buffers, outside the timed region
    copy_gbs = None
    if rank == 0:
        src = torch.empty(out.numel() * out.element_size() // 4, dtype=torch.float32, device=device).normal_()
        dst = torch.empty_like(src)
        dst.copy_(src); torch.cuda.synchronize()
        c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        c0.record()
        for _ in range(10):
            dst.copy_(src)
        c1.record(); torch.cuda.synchronize()
        copy_gbs = 2 * src.numel() * 4 * 10 / (c0.elapsed_time(c1) * 1e-3) / 1e9
        del src, dst
    if plan_cnt is not None:
        plan_cnt()                                          # statistics pass, outside the timed region
    torch.cuda.synchronize()
    mot.check_status()

    tokens_per_step = B * T
    kernel_ms_ranks = [kernel_ms]
    if use_dist:
        import torch.distributed as dist
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        counters = counters.to(cdev)
        dist.all_reduce(counters, op=dist.ReduceOp.SUM)     # the path's only collective: <= 64 B over RCCL/xGMI
        kall = torch.zeros(world, dtype=torch.float64, device=cdev)
        kall[rank] = kernel_ms
        dist.all_reduce(kall, op=dist.ReduceOp.SUM)
        kernel_ms_ranks = kall.tolist()
        elapsed, kernel_ms = float(tmax.item()), max(kernel_ms_ranks)
    total_tokens = tokens_per_step * world * args.steps

    if rank == 0:
        bpt_alg = algorithmic_bytes_per_token(wl, args.ids, esize)
        launch_bytes = bpt_alg * tokens_per_step
        achieved = launch_bytes / (kernel_ms * 1e-3) / 1e9
        traffic, traffic_note = None, "no PMC record for this workload"
        tfile = REPO / "profiles" / "traffic.json"
        if tfile.exists() and world == 1:
            try:
                tj = json.loads(tfile.read_text())
                key = f"{wl}_{args.ids}" + ("" if args.dtype == "f32" else "_bf16") + ("_uniform" if args.uniform_ids else "")
                det = tj.get(key + "_detail", {})
                if key in tj and det.get("source_sha16") == source_sha16():
                    traffic, traffic_note = tj[key], f"rocprofv3 --pmc passes, {det.get('profile')}, kernel sources {det.get('source_sha16')}"
                elif key in tj:
                    traffic_note = (f"profiles/traffic.json[{key}] was measured on kernel sources {det.get('source_sha16')}, the tree is at "
                                    f"{source_sha16()}: refused as stale (re-run tools/pmc_traffic.sh)")
            except Exception as exc:    # a malformed file is not a reason to lose the bench line
                traffic_note = f"profiles/traffic.json unreadable: {exc}"
        c = counters.tolist()
        res = {
            "metric": "mixed-embed tokens/sec at BxT=256x2048; achieved HBM GB/s vs peak",
            "value": total_tokens / elapsed,
            "unit": "tokens/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": "f32" if args.dtype == "f32" else "bf16 tables/output, f32 arithmetic", "data": "synthetic",
            "config": {"workload": f"{wl}: BxT={B}x{T} per GPU"
                                   + (f" (rows {row0}..{row0 + B - 1} of one {B * world}x{T} batch)" if args.scaling == "strong" and world > 1 else "")
                                   + f", vocab {vocab}, bpt {bpt}, d_model {D}, byte_dim {Db}, "
                                   f"mode {mode}{'+rmsnorm' if mode != 'mean' else ''}, ids {args.ids}, token ids "
                                   f"{'uniform' if args.uniform_ids or mode == 'mean' else 'FineWeb-shaped (u^3 skew, EOT p=1/700)'}, "
                                   f"ttb {inp['ttb_kind']}",
                       "global_tokens_per_step": tokens_per_step * world, "per_gpu_tokens": tokens_per_step,
                       "parallelism": f"batch-sharded x{world}"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_note": traffic_note,
                         "kernel": "embed_mean_lds_kernel" if mode == "mean" else "embed_mix_kernel", "kernel_ms": kernel_ms, "launch_us": launch_us,
                         "per": "rank (slowest rank's launch time; every rank moves the same algorithmic bytes)",
                         "kernel_ms_per_rank": kernel_ms_ranks, "aggregate_GBps": achieved * world,
                         "device_copy_GBps": copy_gbs, "frac_of_device_copy": (achieved / copy_gbs) if copy_gbs else None,
                         "algorithmic_bytes_per_token": bpt_alg, "tokens_per_launch": tokens_per_step,
                         "single_gpu_shard_reference": shard_reference(wl, args, tokens_per_step)},
            **({"rehearsal": True, "rehearsal_note": "all ranks on cuda:0, collectives over gloo: NOT a multi-GPU measurement"}
               if args.one_device else {}),
            "library": mot.build_info(),
            "byte_stats": {"tokens": c[0], "slots": c[1], "pads_before": c[2], "pads_after": c[3],
                           "mean_valid_per_token": (c[1] - c[2]) / max(c[0], 1),
                           "pulled_fill": (c[2] - c[3]) / max(c[1], 1)},
        }
        if mode == "concat_linear":   # dense contraction: priced against the fp32 matrix peak, not HBM
            K = WORKLOADS[wl][8] + bpt * Db
            tf = 2.0 * K * D * tokens_per_step / (kernel_ms * 1e-3) / 1e12
            peak = F32_MFMA_PEAK_TFLOPS if args.dtype == "f32" else 2500.0   # dense bf16 MFMA peak, MI355X_MICROARCH.md
            res["roofline"] = {"bound": "mfma", "achieved": tf, "peak": peak, "unit": "TFLOP/s",
                               "frac": tf / peak, "traffic": traffic,
                               "kernel": (("embed_mix_linear_kernel" if args.dtype == "f32" else "embed_mix_linear_bf16_kernel")
                                          if os.environ.get("MOT_LIN_FUSED") else
                                          ("wave_ids16_kernel + " if args.ids == "fused" else "") + "concat16_gemm_kernel (whole call)"
                                          if args.dtype == "bf16" and not os.environ.get("MOT_LIN_COMPOSED") else
                                          ("tokens_to_bytes + pull_bytes + " if args.ids == "fused" else "") + "concat_rows + " +
                                          ("gemm_rows_bt_kernel" if args.dtype == "f32" else "gemm_rows_bf16_kernel") +
                                          " + rows_rms_inplace (whole call)"),
                               "kernel_ms": kernel_ms, "flop_per_token": 2 * K * D, "tokens_per_launch": tokens_per_step}
        # the backward record: on request, and with the headline line itself (one GPU, default workload, secondary records not
        # switched off) so that the driver's own run carries it
        want_bwd = args.backward or (world == 1 and wl == "c4" and args.ids == "fused" and not args.uniform_ids and not args.no_extra)
        if want_bwd and mode == "sum":
            from mixture_of_tokenizers_amd import data_creation as dc
            F = mot.functional
            ids_b = dc.pull_from_left(dc.tokens_to_bytes(toks, tab), bpt, 456, 457)
            gout = torch.randn_like(out)
            into = {"tok_table": torch.zeros_like(inp["tok_table"], dtype=torch.float32),     # gradients are fp32 for bf16 tables too
                    "byte_table": torch.zeros_like(inp["byte_table"], dtype=torch.float32)}
            bkw = dict(mode="sum", bpt=bpt, ids_a=ids_b, norm_out=True, into=into)
            nb = max(1, args.steps // 4)
            # (a) as the autograd node runs it: the positions were grouped by token beside the forward (mot_token_order on a side
            #     stream), the backward call is the scatter kernel alone; (b) the grouping by itself; (c) a backward that groups
            #     the positions itself (round 2's figure); (d) forward + grouping on two streams against the forward alone
            order = F.token_order(toks, inp["tok_table"].shape[0])
            bms = timed_launches(lambda: F.embed_mix_backward(gout, toks, inp["tok_table"], inp["byte_table"], token_order=order, **bkw), nb, warm=3)
            oms = timed_launches(lambda: F.token_order(toks, inp["tok_table"].shape[0]), nb, warm=3)
            sms = timed_launches(lambda: F.embed_mix_backward(gout, toks, inp["tok_table"], inp["byte_table"], **bkw), nb, warm=3)
            side = torch.cuda.Stream(device=device)

            def fwd_and_order():
                side.wait_stream(torch.cuda.current_stream(device))
                with torch.cuda.stream(side):
                    F.token_order(toks, inp["tok_table"].shape[0])
                plan()
                torch.cuda.current_stream(device).wait_stream(side)

            fms = timed_launches(plan, nb, warm=3)
            foms = timed_launches(fwd_and_order, nb, warm=3)
            read_bytes = 2 * out.element_size() * D * tokens_per_step   # grad_out row + token row per position (byte rows and ids come from L2)
            kname = "embed_mix_bwd_plain_kernel"
            res["backward"] = {"kernel": kname, "kernel_ms": bms, "tokens_per_s": tokens_per_step / (bms * 1e-3),
                               "hbm_read_GBps": read_bytes / (bms * 1e-3) / 1e9, "hbm_peak_GBps": HBM_PEAK_GBS,
                               "token_order_ms": oms, "backward_grouping_itself_ms": sms,
                               "forward_ms": fms, "forward_plus_token_order_on_a_side_stream_ms": foms,
                               "note": "kernel_ms: the backward call with the positions already grouped by token (mot_token_order: a counting "
                                       "sort, bwd_rank / bwd_scan / bwd_place, run once per batch beside the forward) = one scatter kernel: a "
                                       "token row is read and its gradient row flushed (fp32 atomic row-add) once per run, byte-table "
                                       "gradient in 64-bit fixed point in LDS; backward_grouping_itself_ms = the same call without a given "
                                       "order; hbm_read counts the algorithmic grad_out row + token row per position"}
        if world == 1 and mode == "sum" and wl == "c4" and args.ids == "fused" and not args.uniform_ids and not args.no_extra:
            del plan, plan_cnt, out
            torch.cuda.empty_cache()
            res["extra"] = extra_records(mot, device, args.dtype, args.steps, args.warmup)
        if not args.no_cpu_baseline and world == 1:
            res["cpu_baseline"] = cpu_baseline(wl, inp, args.cpu_seconds)
        print(json.dumps(res), flush=True)
    if use_dist:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
else:
    _imports()   # imported as a module (tools/bench_*.py): the helpers above need numpy and torch bound
