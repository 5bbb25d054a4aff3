#!/usr/bin/env python
"""bench.py - tri-modal images/sec, forward+backward+optimizer step, of the EDITOR hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--dtype bf16|f16|f16x2|f32] [--preset RGBNT201|RGBNT100|MSVR310|SYNTH4L]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]): RGBNT201 cfg, 3 modalities, ViT-B/16, 256x128, B=128 PER GPU (weak scaling, SURVEY.md 7
"DDP batch semantics"), bf16 MFMA with fp32 accumulation/residual/grads, synthetic seeded uint8-derived images, random-init
weights of the real architecture.  A "step" is what engine/processor.py:70-107 does per batch:
zero_grad -> forward -> loss (pairs + aux) -> backward (+ gradient all-reduce, overlapped) -> SGD step.

`value` (the headline) is the reference's own throughput definition (engine/processor.py:107,114-118): B / mean
seconds per iteration of a loop that hands every step its batch from pinned host memory (processor.py:73-78; the copy of
batch i+1 is pipelined under step i, as a prefetching loader does) and synchronises the device every iteration; K iterations
between a barrier + synchronize on both sides, MAX over ranks.  `replay_only`: the bare step beside it (inputs resident,
one synchronisation at the end; N = 1 only).

One JSON line on rank 0.
  roofline      the dominant kernel family (16-bit MFMA GEMM): algorithmic FLOPs of every launch of one step / their
                HIP-event durations (replayed back to back after the timed region), against the 2.5 PFLOP/s dense peak;
                `traffic` = HBM bytes per step of those launches from the committed PMC profile of the same command
                (the newest profiles/rNN_pmc_traffic.json, tools/pmc_traffic.sh; `traffic_source` names it - PMC counters need their
                own rocprofv3 passes and are not collected by this run), null when absent; `step_frac` = SURVEY.md 8(d)'s
                algorithmic FLOPs of the WHOLE step / ms_per_step / peak;
                `hbm_kernels` = the memory-bound select / gather / normalisation kernels timed live with HIP events
                against the 8 TB/s HBM peak (algorithmic bytes).
  eval          forward-only throughput of model.eval() at the same batch (do_inference, engine/processor.py:217-270)
  modes         every compute mode beside the benchmarked one - bf16, f16, f16x2 (split-precision forward), f32 (exact-f32
                parity mode): images/sec of the same timed loop (child process each) AND its accuracy against the oracle
                (cls4t relative error, token-selection agreement), so the speed is never read without the parity it buys.
  other_configs BASELINE.json configs 3 / 4 / 5 (RGBNT100, MSVR310 384x128, the synthetic 4-modal ViT-L) on this one GPU: the same timed loop
                in a child process per preset - per-GPU workloads of configs quoted on 8 GPUs, so that the driver's own run carries
                a number for each; not part of `value`.  N = 1 default line only (--no-others skips it).
  cpu_baseline  the oracle (CPU restatement pinned to the reference) timed on this host's cores: the same step at B=128
                (1 warm-up + 3 timed iterations: forward, the reference's loss, backward, SGD) and config 1 (`c1`: B=32,
                RGB only, backbone forward).  N = 1 only.
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_TFLOPS = 2500.0               # MI355X dense bf16 / f16 MFMA (MI355X_MICROARCH.md)
PEAK_F32_TFLOPS = 157.3            # dense fp32 MFMA (v_mfma_f32_16x16x4_f32), parity mode
PEAK_HBM_GBS = 8000.0              # HBM3E


class _Writer:                     # engine/processor.py:42 passes a SummaryWriter into forward
    def add_scalar(self, tag, value, step=None):
        self.last = value          # no host sync in the timed region


class _GemmProbe:
    """HIP-event timing of the dominant kernel family (16-bit MFMA GEMM).

    Every 16-bit GEMM launch of ONE step is recorded (arguments kept alive); after the timed region the recorded
    launches are replayed back to back on the same stream between two HIP events.  Timing each launch in place would
    fold host launch gaps of the eager step into the kernel time (measured: +17 %), which is not a property of the
    kernel; the replay has the same operands, shapes and epilogues and no other work in between.  The rocprofv3
    --kernel-trace --stats summary of the same command (profiles/) gives the in-situ durations and agrees."""

    def __init__(self):
        self.calls = []
        self.recording = False
        self.timing = None          # list while ONE eager step is timed launch by launch, in place (in_situ below)
        self.entry_timing = None    # list of (C-ABI entry name, e0, e1) while every library call of an eager step is timed in place
        self.entries = {}           # entry name -> (ms per step, calls per step): the non-GEMM kernels INSIDE the step, this run

    def _timed(self, kind, work, fn):
        """fn() between two HIP events on the CURRENT stream (= the stream the launch goes to: the wrappers run inside whatever
        stream context the caller set)."""
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r = fn()
        e1.record()
        self.timing.append((kind, work, e0, e1))
        return r

    def install(self):
        from editor_amd import ops
        self._orig = ops.gemm
        probe = self

        def recorded(a, b, c, m, n, k, *args, **kw):
            if probe.recording and a.dtype in (torch.bfloat16, torch.float16):
                ta = args[3] if len(args) > 3 else kw.get("trans_a", 0)
                tb = args[4] if len(args) > 4 else kw.get("trans_b", 0)
                kind = kw.get("tag") or ("fwd" if not ta and not tb else ("dgrad" if not ta else "wgrad"))
                # compacted HMA launches are sized for the worst case; count only the live rows (device scalar,
                # read after the timed region) as algorithmic work
                probe.calls.append([kind, probe._work(m, n, k, kw, ta), (a, b, c, m, n, k) + args, dict(kw)])
            if probe.timing is not None and a.dtype in (torch.bfloat16, torch.float16):
                ta = args[3] if len(args) > 3 else kw.get("trans_a", 0)
                tb = args[4] if len(args) > 4 else kw.get("trans_b", 0)
                kind = kw.get("tag") or ("fwd" if not ta and not tb else ("dgrad" if not ta else "wgrad"))
                return probe._timed(kind, probe._work(m, n, k, kw, ta), lambda: probe._orig(a, b, c, m, n, k, *args, **kw))
            return probe._orig(a, b, c, m, n, k, *args, **kw)
        ops.gemm = recorded
        # the HMA head's per-modality blocks leave as grouped launches (editor_gemm_group): `cnt` products of identical shape
        self._orig_fgroup = ops.gemm_group

        def recorded_fgroup(reqs):
            a0, kw0 = reqs[0]
            m, n, k = a0[3:6]
            kind = kw0.get("tag") or "fwd"
            if probe.recording:
                probe.calls.append([kind, (m * len(reqs), n, k, kw0.get("m_live"), False, len(reqs)), ("fgroup", list(reqs)), {}])
            if probe.timing is not None:
                return probe._timed(kind, (m * len(reqs), n, k, kw0.get("m_live"), False, len(reqs)), lambda: probe._orig_fgroup(reqs))
            return probe._orig_fgroup(reqs)
        ops.gemm_group = recorded_fgroup
        # split-precision forward products ('f16x2'): algorithmic FLOPs 2 m n k (the three MFMA passes are the price of
        # fp32-class products on the half matrix cores, not extra algorithmic work)
        self._orig_split = ops.gemm_split

        def recorded_split(a, b, c, c_lo, m, n, k, **kw):
            if probe.recording:
                probe.calls.append(["fwd", probe._work(m, n, k, kw, False), ("split", a, b, c, c_lo, m, n, k), dict(kw)])
            if probe.timing is not None:
                return probe._timed("fwd", probe._work(m, n, k, kw, False), lambda: probe._orig_split(a, b, c, c_lo, m, n, k, **kw))
            return probe._orig_split(a, b, c, c_lo, m, n, k, **kw)
        ops.gemm_split = recorded_split
        # the four weight gradients of a block as one grouped launch (editor_gemm_wgrad_group)
        self._orig_group = ops.gemm_wgrad_group

        def recorded_group(jobs, m, alpha=1.0, m_live=None):
            # (m, n, k) of the probe's FLOP formula 2 m n k with the token rows as the reduction: n * k -> sum_i N_i K_i; a problem
            # with its own live count (4th entry: stochastic-depth-compacted rows) EXECUTES only that many reduction rows
            nk = sum(j[0].shape[1] * j[1].shape[1] for j in jobs)
            skip = [(j[0].shape[1] * j[1].shape[1], j[3] if len(j) > 3 else None) for j in jobs]
            work = (nk, 1, m, m_live, True, 1, skip if any(l_ is not None for _, l_ in skip) else None)
            if probe.recording:
                probe.calls.append(["wgrad", work, ("group", list(jobs), m, alpha, m_live), {}])
            if probe.timing is not None:
                return probe._timed("wgrad", work, lambda: probe._orig_group(jobs, m, alpha, m_live))
            return probe._orig_group(jobs, m, alpha, m_live)
        ops.gemm_wgrad_group = recorded_group

    def remove(self):
        from editor_amd import ops
        ops.gemm = self._orig
        ops.gemm_split = self._orig_split
        ops.gemm_wgrad_group = self._orig_group
        ops.gemm_group = self._orig_fgroup

    def _run(self, args, kw):
        if args and isinstance(args[0], str):
            if args[0] == "group":
                return self._orig_group(*args[1:], **kw)
            if args[0] == "fgroup":
                return self._orig_fgroup(args[1])
            return self._orig_split(*args[1:], **kw)
        return self._orig(*args, **kw)

    @staticmethod
    def _work(m, n, k, kw, ta):
        """(m, n, k, live, trans_a, cnt, skip): `live` = the compacted HMA head's live rows (its algorithmic work IS the
        kept tokens'); `skip` = the live prefix of stochastic-depth-compacted rows (kw live_dense: the reference computes the dropped
        samples' branch and multiplies it by 0 - algorithmic work stays the full m, EXECUTED work is the prefix)."""
        dense = bool(kw.get("live_dense"))
        return (m, n, k, None if dense else kw.get("m_live"), bool(ta), 1, kw.get("m_live") if dense else None)

    @staticmethod
    def _flops(work, executed=False):
        """algorithmic FLOPs 2 m n k of a launch with its LIVE row count (compacted HMA launches are sized for the worst case;
        a grouped launch of cnt products shares one live count per product).  executed=True: what the launch really multiplied -
        minus the rows stochastic depth dropped (see _work)."""
        m, n, k, live, ta = work[:5]
        cnt = work[5] if len(work) > 5 else 1
        skip = work[6] if len(work) > 6 else None
        if live is not None:
            rows = int(live.item())
            if ta:
                k = min(k, rows)
            else:
                m = min(m, rows * cnt)
        if executed and skip is not None:
            if isinstance(skip, list):                       # grouped weight gradients: (N_i K_i, live_i) per problem
                return sum(2.0 * nk * (k if l_ is None else min(k, int(l_.item()))) for nk, l_ in skip)
            m = min(m, int(skip.item()))
        return 2.0 * m * n * k

    def in_situ(self, step_fn):
        """Eager steps (one untimed, two timed: per launch the MEAN of its two timings - VERDICT r5 item 8) with every 16-bit GEMM launch (incl. a weight-gradient launch's slab reduction) bracketed by HIP events on
        the stream it runs on, the weight gradients on the MAIN stream (so that no launch shares the chip with another and the
        durations add up: the `serial` profile's condition, profiles/rNN_bench_kernel_stats_serial.csv) -> {kind: (flops, ms, n)}.
        The operands are the step's own, in the cache state the step leaves them in - the figure the rocprofv3 kernel-trace summary
        of the same command must agree with."""
        from editor_amd import functional as fn
        from editor_amd import ops
        side, fn.WGRAD_SIDE_STREAM = fn.WGRAD_SIDE_STREAM, False
        runs = []
        orig_call = ops.call
        probe = self

        def timed_call(name, *a):
            # every C-ABI entry of the library, bracketed like the GEMM launches: the durations the memory-bound kernels have INSIDE
            # this run's step (not a committed builder CSV).  An entry = its launches (attention_bwd: both passes).
            if probe.entry_timing is None:
                return orig_call(name, *a)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            r = orig_call(name, *a)
            e1.record()
            probe.entry_timing.append((name, e0, e1))
            return r
        try:
            step_fn()                       # untimed: the eager step's allocations come out of the caching allocator afterwards (after a
            torch.cuda.synchronize()        # hipGraph capture the ordinary pool is empty: a starved GPU puts host latency between the events)
            for _ in range(2):
                self.timing = []
                step_fn()
                torch.cuda.synchronize()
                runs.append(self.timing)
            self.timing = None
            ops.call = timed_call
            ent_runs = []
            for _ in range(2):
                self.entry_timing = []
                step_fn()
                torch.cuda.synchronize()
                ent_runs.append(self.entry_timing)
            self.entry_timing = None
            acc = {}
            for run in ent_runs:
                for name, e0, e1 in run:
                    ms, n = acc.get(name, (0.0, 0))
                    acc[name] = (ms + e0.elapsed_time(e1), n + 1)
            self.entries = {k: (ms / len(ent_runs), n / len(ent_runs)) for k, (ms, n) in acc.items()}
        finally:
            ops.call = orig_call
            fn.WGRAD_SIDE_STREAM = side
            self.timing = None
            self.entry_timing = None
        by_kind = {}
        if len(runs) == 2 and len(runs[0]) == len(runs[1]):
            for (kind, work, e0, e1), (_, _, f0, f1) in zip(*runs):             # per launch: the mean of its two timings
                f, ms, n, fx = by_kind.get(kind, (0.0, 0.0, 0, 0.0))
                by_kind[kind] = (f + self._flops(work), ms + 0.5 * (e0.elapsed_time(e1) + f0.elapsed_time(f1)), n + 1,
                                 fx + self._flops(work, executed=True))
        return by_kind

    def replay(self, reps=3):
        by_kind = {}
        for call in self.calls:                            # algorithmic FLOPs with the live row count
            call[1] = self._flops(call[1])
        for kind in ("fwd", "dgrad", "wgrad"):
            calls = [c for c in self.calls if c[0] == kind]
            if not calls:
                continue
            for _, _, args, kw in calls:                   # warm
                self._run(args, kw)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                for _, _, args, kw in calls:
                    self._run(args, kw)
            e1.record()
            torch.cuda.synchronize()
            by_kind[kind] = (sum(c[1] for c in calls), e0.elapsed_time(e1) / reps, len(calls))
        return by_kind


def _event_us(fn, nsets=1, reps=24, warm=None):
    """Average HIP-event duration of fn(i), i cycling over `nsets` operand sets (the caller sizes the sets so that their
    footprint exceeds 512 MB: the 256 MB Infinity Cache cannot serve a set that was last touched nsets - 1 launches ago)."""
    for i in range(warm if warm is not None else nsets):
        fn(i % nsets)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps):
        fn(i % nsets)
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / reps


def source_hash():
    """sha1 over the sources a kernel duration depends on (csrc, the host modules): stamped beside a committed rocprof summary by
    tools/prof.sh, compared by _in_situ_us - a profile of an OLDER tree is not quoted (the GPU box has no .git)."""
    import glob
    import hashlib
    hsh = hashlib.sha1()
    pats = ("editor_amd/csrc/*.hip", "editor_amd/csrc/*.h", "include/*.h", "editor_amd/*.py", "editor_amd/modeling/*.py")
    for f in sorted(f for p_ in pats for f in glob.glob(os.path.join(ROOT, p_))):
        hsh.update(os.path.relpath(f, ROOT).encode())
        hsh.update(open(f, "rb").read())
    return hsh.hexdigest()[:16]


def _in_situ_us():
    """{kernel-name prefix: average us} from the newest committed rocprofv3 --kernel-trace --stats summary of the bench command
    with the side stream off (profiles/rNN_bench_kernel_stats_serial.csv): the durations the kernels have INSIDE the step, on
    operands the previous kernels pushed out of the caches."""
    import csv
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_bench_kernel_stats_serial.csv")))
    if not files:
        return None, {}
    stamp = files[-1][:-4] + ".hash"
    if not os.path.exists(stamp) or open(stamp).read().strip() != source_hash():
        return os.path.relpath(files[-1], ROOT) + " (NOT quoted: the profile was taken on a different source tree)", {}
    out = {}
    try:
        for r in csv.DictReader(open(files[-1])):
            out[r["Name"].replace("void ", "").replace("(anonymous namespace)::", "")] = float(r["AverageNs"]) / 1e3
    except Exception:
        return None, {}
    return os.path.relpath(files[-1], ROOT), out


def hbm_kernels(model, img, b, act_dtype, entries=None):
    """The memory-bound kernels of the path (north_star: "achieved HBM GB/s for the memory-bound select/gather"), each
    timed live on this GPU with HIP events on its real operand shapes; algorithmic bytes per launch from SURVEY.md 8(d).
    Every kernel rotates over `sets` independent operand sets whose footprint exceeds 512 MB, so the figure is HBM, not the
    256 MB Infinity Cache (round 3 repeated one set: freq_counts read 0.47 there and 0.29 inside the step).  `in_situ_us` is
    the same kernel's average duration in the committed rocprofv3 profile of the whole step."""
    from editor_amd import ops
    dev = img[next(iter(img))].device
    base = model.BACKBONE.base
    nmod, d, heads = model.nmod, base.embed_dim, base.heads
    t = base.num_patches + 1
    m = nmod * b * t
    mods = [v for v in img.values()]
    out = []
    # in_situ_*: the same kernel INSIDE the step.  This run's own figures (entries: C-ABI entry -> (ms per step, calls per step), HIP
    # events around every library call of two eager steps, _GemmProbe.in_situ) when available; else the committed rocprofv3 CSV.
    src, situ = _in_situ_us()
    if entries:
        src, situ = "this run: HIP events around every library call of two eager steps (weight gradients on the main stream)", {}

    def nsets_for(footprint):
        return max(3, int(600e6 // max(footprint, 1)) + 1)

    def add(name, alg_bytes, fn, nsets, match, situ_bytes=None, situ_note=None, entry=None, entry_calls=None):
        us = _event_us(fn, nsets)
        gbs = alg_bytes / us / 1e3
        ent = {"kernel": name, "alg_bytes": int(alg_bytes), "us": round(us, 1), "GB/s": round(gbs, 0),
               "frac": round(gbs / PEAK_HBM_GBS, 3), "sets": nsets}
        hit = [v for k, v in situ.items() if all(tok in k for tok in match)]
        if entries and entry:
            # the entry's launches of full size (entry_calls per step: the backbone's; the HMA head's few small ones ride along in
            # the total - a slight over-estimate of the per-launch time)
            tot = [(ms, n) for k, (ms, n) in entries.items() if (k == entry[:-1] if entry.endswith("$") else k.startswith(entry))]
            if tot:
                ms_, n_ = sum(t_[0] for t_ in tot), sum(t_[1] for t_ in tot)
                hit = [1e3 * ms_ / (entry_calls or n_)]
                ent["in_situ_calls_per_step"] = round(n_, 1)
        if hit:
            ent["in_situ_us"] = round(max(hit), 1)
            ent["in_situ_frac"] = round((situ_bytes or alg_bytes) / max(hit) / 1e3 / PEAK_HBM_GBS, 3)
            if situ_note:
                ent["in_situ_note"] = situ_note
        out.append(ent)

    px = mods[0].numel() * 4
    n = nsets_for(nmod * px)
    imgs = [[v.clone() for v in mods] for _ in range(n)]
    add("freq_counts4_kernel (Haar DWT -> mean -> IDWT -> positive count)", nmod * px,
        lambda i: ops.freq_counts(imgs[i][0], imgs[i][1], imgs[i][2], imgs[i][3] if nmod > 3 else None), n, ("freq_counts",),
        entry="editor_freq_counts")
    del imgs
    n = nsets_for(2 * nmod * b * t * d * 4)
    feats = [torch.randn(nmod, b, t, d, device=dev) for _ in range(n)]
    index = (torch.rand(b, t - 1, device=dev) > 0.5).to(torch.uint8)
    add("sfts_apply_kernel (mask apply + BCC partial sums)", 2 * feats[0].numel() * 4, lambda i: ops.sfts_apply(feats[i], index, True),
        n, ("sfts_apply_kernel",), entry="editor_sfts_apply$")
    del feats
    g = torch.ones(d, device=dev)
    bb = torch.zeros(d, device=dev)
    if act_dtype != torch.float32:
        n = nsets_for(m * 3 * d * 2)
        qkvs = [(torch.randn(m, 3 * d, device=dev) * 0.5).to(act_dtype) for _ in range(n)]
        lses = [ops.attention_fwd(q, nmod * b, t, heads, d // heads)[1] for q in qkvs]
        add("attn_rollout_step_kernel (one layer: q,k + lse in, r out)", m * 2 * d * 2 + 2 * heads * m * 4,
            lambda i: ops.attn_rollout_qk([(qkvs[i], lses[i])], nmod * b, t, heads, d // heads), n, ("attn_rollout_step_kernel",),
            entry="editor_attn_rollout_step")
        add("attn_q_pass_kernel fwd (qkv in, o out)", m * 4 * d * 2,
            lambda i: ops.attention_fwd(qkvs[i], nmod * b, t, heads, d // heads), n, ("attn_q_pass_kernel", "false, true, false, false"),
            entry="editor_attention_fwd", entry_calls=base.depth)
        del qkvs, lses
    esz = 4 if act_dtype == torch.float32 else 2
    n = nsets_for(m * d * (4 + esz))
    xs = [torch.randn(m, d, device=dev) for _ in range(n)]
    add("layernorm_fwd_kernel", m * d * (4 + esz), lambda i: ops.layernorm_fwd(xs[i], g, bb, 1e-6, act_dtype), n,
        ("layernorm_fwd_kernel", "unsigned short" if esz == 2 else "float"), entry="editor_layernorm_fwd",
        entry_calls=2 * base.depth + 1)
    n = 3
    xs = xs[:n]
    ys, means, rstds = zip(*[ops.layernorm_fwd(x, g, bb, 1e-6, act_dtype) for x in xs])
    dxs = [torch.randn(m, d, device=dev) for _ in range(n)]
    add("layernorm_bwd_kernel (+ residual-gradient add)", m * d * (esz + 4 + 4 + 4),
        lambda i: ops.layernorm_bwd(ys[i], xs[i], g, means[i], rstds[i], dx_in=dxs[i]), n,
        ("layernorm_bwd_kernel", "unsigned short, 3, true" if esz == 2 else "float"),
        situ_bytes=m * d * (esz + 4 + 4 + 4 + esz) if esz == 2 else None,
        situ_note="in the step the M = 3*B*T launches are the CAST form (also writes the 16-bit copy of dx: + M*D*2 bytes); the plain "
                  "instantiation of the profile is the compacted HMA head's small launches" if esz == 2 else None,
        entry="editor_layernorm_bwd_cast" if esz == 2 else "editor_layernorm_bwd", entry_calls=2 * base.depth - 1 if esz == 2 else None)
    return out, src


def mode_accuracy(preset, dtypes, batch=16, seed=31):
    """Accuracy of every compute mode beside its speed: eval forward of `batch` tri-modal samples against the oracle
    (oracle/editor_ref.py, pinned to the reference by tests/golden) - selection agreement free-running, `cls4t` relative
    error with the oracle's selection teacher-forced (the protocol of tests/test_gpu_fullsize.py; north_star: indices
    bit-identical, features within 1e-3)."""
    import contextlib
    import io
    from oracle import editor_ref as oracle
    from editor_amd import config, synth
    from editor_amd.modeling import make_model
    torch.set_num_threads(_usable_cores())
    cfg, c, cams = config.preset(preset, drop_path=0.0)
    with contextlib.redirect_stdout(io.StringIO()):
        m0 = make_model(cfg, c, cams)
    synth.fill_state_dict_(m0.state_dict(), seed)
    sd = {k: v.clone() for k, v in m0.state_dict().items()}
    h, w = cfg.INPUT.SIZE_TRAIN
    img, label, cam, view = synth.make_batch(seed + 1, batch, h, w, cams, instances=2)
    with torch.no_grad():
        ref, aux = oracle.editor_forward(sd, img, cam, training=False, al=cfg.MODEL.AL, return_aux=True)
    gimg = {k: v.cuda() for k, v in img.items()}
    res = {}
    for dt in dtypes:
        cfg_d, c, cams = config.preset(preset, compute_dtype=dt, drop_path=0.0)
        with contextlib.redirect_stdout(io.StringIO()):
            m = make_model(cfg_d, c, cams)
        m.load_state_dict(sd)
        m = m.cuda().eval()
        with torch.no_grad():
            m(gimg, cam_label=cam.cuda(), view_label=view.cuda())
            got = m.last_aux
            rows = [(got["attn_masks"][i].cpu().bool() == aux["attn_masks"][i]).all(dim=1).float().mean().item() for i in range(3)]
            same = torch.equal(got["index"].cpu().bool(), aux["index"])
            m.teacher_index = aux["index"]
            out = m(gimg, cam_label=cam.cuda(), view_label=view.cuda())
        err = ((out.cpu().double() - ref.double()).norm() / ref.double().norm()).item()
        res[dt] = {"cls4t_rel_err": float("%.3g" % err), "selection_rows_identical": round(min(rows), 4),
                   "index_bit_identical": bool(same)}
        del m
        torch.cuda.empty_cache()
    return res, batch


def modes_block(args, cfg, cams, own_value, own_eval=None):
    """{mode: img/s (the same timed loop, in a child process per mode) + accuracy vs the oracle}."""
    import subprocess
    modes = ["bf16", "f16", "f16x2", "f16x2s", "f32"]
    acc, nb = mode_accuracy(args.preset, modes)
    out = {"accuracy_sample": f"eval forward, B={nb}, {args.preset}, vs the oracle on the host cores; cls4t with the oracle's "
                              "selection teacher-forced, selection free-running (rows = (sample, modality) attention masks)"}
    for dt in modes:
        entry = dict(acc[dt])
        if dt == args.dtype:
            entry["value"] = own_value
            if own_eval is not None:
                entry["eval"] = own_eval.get("value")
        else:
            steps, warm = (3, 1) if dt == "f32" else (args.steps, args.warmup)
            cmd = [sys.executable, os.path.abspath(__file__), "--dtype", dt, "--preset", args.preset, "--batch", str(args.batch),
                   "--steps", str(steps), "--warmup", str(warm), "--graph", "--no-cpu-baseline", "--no-replay", "--no-modes"]
            if args.no_h2d:
                cmd.append("--no-h2d")
            try:
                cp = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
                lines = [ln for ln in cp.stdout.splitlines() if ln.startswith("{") and '"metric"' in ln]
                j = json.loads(lines[-1])
                entry["value"] = j["value"]
                entry["ms_per_step"] = j["ms_per_step"]
                if "eval" in j:
                    entry["eval"] = j["eval"].get("value")
            except Exception as e:                       # a mode that fails to run is reported, not hidden
                entry["value"] = None
                entry["error"] = type(e).__name__
        # BASELINE.json north_star: "bit-exactly for index/top-k results and within 1e-3 rel for bf16 features"
        entry["meets_north_star"] = {"indices": bool(entry["index_bit_identical"]), "features": entry["cls4t_rel_err"] <= 1e-3}
        out[dt] = entry
    ok = [(out[dt]["value"], dt) for dt in modes if out[dt].get("value") and all(out[dt]["meets_north_star"].values())]
    out["value_at_parity"] = None if not ok else {"mode": max(ok)[1], "value": max(ok)[0], "what": "fastest mode whose selected-token "
                                                  "indices are bit-identical to the oracle's AND whose features are within 1e-3"}
    return out


def other_configs_block(args):
    """BASELINE.json configs 3 / 4 / 5 on this one GPU (they had builder-run lines only): the same timed loop of
    this file in a child process per preset, so that the driver's own run carries a number for each.  They are per-GPU workloads of
    configs the reference quotes on 8 GPUs - not the headline, not part of `value`."""
    import subprocess
    out = {"what": "the same timed loop (hipGraph replay, H2D per batch, sync per iteration) per BASELINE preset, one child process each, "
                   "this GPU; per-GPU batch as the preset's default"}
    for preset, label in (("RGBNT100", "config 3"), ("MSVR310", "config 4"), ("SYNTH4L", "config 5")):
        cmd = [sys.executable, os.path.abspath(__file__), "--dtype", args.dtype, "--preset", preset, "--steps", str(args.steps),
               "--warmup", str(args.warmup), "--graph", "--no-cpu-baseline", "--no-replay", "--no-modes", "--no-eval", "--no-others"]
        try:
            cp = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
            lines = [ln for ln in cp.stdout.splitlines() if ln.startswith("{") and '"metric"' in ln]
            j = json.loads(lines[-1])
            out[preset] = {"baseline_config": label, "value": j["value"], "unit": j["unit"], "ms_per_step": j["ms_per_step"],
                           "workload": j["config"]["workload"]}
        except Exception as e:                           # reported, not hidden; never fails the headline
            out[preset] = {"baseline_config": label, "value": None, "error": type(e).__name__}
    return out


def shader_clock_mhz(work, busy_ms=60.0, n=3000, sleep=6):
    """The clock the CUs run at while `work` (a callable launching on the current stream) keeps the chip busy: one resident
    wave samples (s_memtime = shader-clock cycles, s_memrealtime = the constant 100 MHz reference) on a second stream
    (csrc/probe.hip: editor_probe_clock_trace; tools/clock_probe.py).  MI355X_MICROARCH.md quotes the MFMA peak at the
    2.4 GHz boost clock and documents that the chip clocks to its power budget under dense GEMM load."""
    import ctypes
    from editor_amd import _lib
    fn = _lib.probe_lib().editor_probe_clock_trace
    fn.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    fn.restype = ctypes.c_int
    buf = torch.zeros(2 * n, dtype=torch.int64, device="cuda")
    s_probe = torch.cuda.Stream()
    work()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    work()
    e1.record()
    torch.cuda.synchronize()
    reps = max(2, int(busy_ms / max(e0.elapsed_time(e1), 1e-3)))
    if fn(buf.data_ptr(), n, sleep, s_probe.cuda_stream) != 0:
        return None
    for _ in range(reps):
        work()
    torch.cuda.synchronize()
    t = buf.view(n, 2).cpu().numpy().astype("float64")
    cyc, ref = t[:, 0] - t[0, 0], (t[:, 1] - t[0, 1]) / 100.0                  # shader cycles, microseconds
    span = min(ref[-1], busy_ms * 1e3)
    i0, i1 = int((ref >= 0.3 * span).argmax()), int((ref >= 0.9 * span).argmax())
    if i1 <= i0:
        return None
    # (mean over the busy window, slowest and fastest 0.5 ms inside it)
    w = 25
    win = [(cyc[j + w] - cyc[j]) / (ref[j + w] - ref[j]) for j in range(i0, i1 - w, w) if ref[j + w] > ref[j]]
    return float((cyc[i1] - cyc[i0]) / (ref[i1] - ref[i0])), (float(min(win)), float(max(win))) if win else None


def eval_throughput(model, img, cam, view, b, steps):
    """Forward-only throughput of the boundary's OTHER caller, do_inference (engine/processor.py:217-270): model.eval(), no_grad,
    `model(img, cam_label=, view_label=)` -> (B, 2304) features, inputs resident, timed as `steps` back-to-back forwards between
    two synchronisations - as a hipGraph replay like the training step, and eagerly (the serial selection kernels are on
    the critical path here: nothing hides them behind a side stream's weight gradients)."""
    was_training = model.training
    model.eval()
    res = {"what": "model.eval() forward at the same batch (do_inference, engine/processor.py:217-270): images/sec, inputs resident"}
    try:
        with torch.no_grad():
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(2):
                    out = model(img, cam_label=cam, view_label=view)
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                out = model(img, cam_label=cam, view_label=view)
            torch.cuda.synchronize()
            el = time.perf_counter() - t0
            res["eager"] = {"value": round(b * steps / el, 1), "ms_per_batch": round(1e3 * el / steps, 3)}
            try:
                torch.cuda.empty_cache()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    out = model(img, cam_label=cam, view_label=view)
                g.replay()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(steps):
                    g.replay()
                torch.cuda.synchronize()
                el = time.perf_counter() - t0
                res["graph"] = {"value": round(b * steps / el, 1), "ms_per_batch": round(1e3 * el / steps, 3)}
                del g
            except Exception as e:
                res["graph"] = {"error": f"{type(e).__name__}: {e}"[:200]}
                torch.cuda.synchronize()
            res["value"] = max(v["value"] for v in (res.get("eager"), res.get("graph")) if v and "value" in v)
            res["features_finite"] = bool(torch.isfinite(out).all())
    except Exception as e:
        res["error"] = f"{type(e).__name__}: {e}"[:300]
    model.train(was_training)
    return res


def torch_gpu_yardstick():
    """Plain PyTorch-ROCm (vendor BLAS + ATen, autocast) on THIS GPU for the dominant part of the step - the 12 ViT-B blocks over
    3 x 128 sequences of 129 tokens, forward + backward + SGD, written as the reference writes them
    (tools/torch_backbone_reference.py, run in a child process after this process has released its memory).  A yardstick for
    the reader, beside `value`, which times the WHOLE step; nothing on the product path uses it."""
    import subprocess
    torch.cuda.empty_cache()
    try:
        cp = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "torch_backbone_reference.py")], stdout=subprocess.PIPE,
                            stderr=subprocess.DEVNULL, text=True, timeout=600)
        ms = {ln.split()[1]: float(ln.split()[2]) for ln in cp.stdout.splitlines() if ln.startswith("YARDSTICK")}
    except Exception as e:
        return {"error": type(e).__name__}
    return {"what": "plain PyTorch-ROCm, autocast: ONLY the 12 ViT-B/16 blocks over the 3 x 128 stacked sequences (explicit q k^T softmax, "
                    "attention maps kept for the rollout, as vit_pytorch.py:184-198,215-220), forward + backward + torch.optim.SGD, same GPU",
            "ms_per_step": ms, "this_repo_whole_step_is": "ms_per_step of this line (patch embedding, blocks, SFTS, HMA, loss head, backward, SGD)"}


def _usable_cores():
    """Cores this process may really use: affinity mask, capped by the cgroup CPU quota (a container that
    reports 256 logical CPUs but is throttled to a few thrashes when handed 256 threads)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, min(n, 32))          # the oracle's per-op parallelism saturates well below 32 threads


def cpu_baseline(model, cfg, cams, batch, iters):
    """SURVEY.md 8(d) "CPU baseline beside it": the oracle (oracle/editor_ref.py, proven equal to the reference by the
    committed fixtures) on this host's cores, fp32, same synthetic batch, same step definition as the timed GPU step:
    forward + the reference's loss (label-smoothed CE + batch-hard triplet over the pairs + aux) + backward + SGD step;
    1 warm-up + `iters` timed iterations at B = `batch`.  Plus config 1: B=32, RGB only, backbone forward."""
    from oracle import editor_ref as oracle
    from editor_amd import synth
    cores = _usable_cores()
    torch.set_num_threads(cores)
    sd = {k: v.detach().float().cpu().clone() for k, v in model.state_dict().items()}
    leaves = []
    for k, v in sd.items():
        if v.is_floating_point() and "centers" not in k and "running" not in k and not k.startswith("FREQ"):
            v.requires_grad_(True)
            leaves.append(v)
    opt = torch.optim.SGD(leaves, lr=1e-3, momentum=0.9, weight_decay=1e-4)
    h, w = cfg.INPUT.SIZE_TRAIN
    nmod = int(getattr(cfg.MODEL, "NUM_MODALITIES", 3))
    mods = oracle.MODALITIES4 if nmod == 4 else oracle.MODALITIES3
    heads = model.BACKBONE.base.heads
    img, label, cam, view = synth.make_batch(1111, batch, h, w, cams, instances=min(16, max(batch // 2, 1)),
                                             keys=[m[0] for m in mods])

    def run():
        t0 = time.perf_counter()
        opt.zero_grad(set_to_none=True)
        out = oracle.editor_forward(sd, img, cam, label=label, training=True, al=cfg.MODEL.AL, heads=heads,
                                    hma_heads=model.hma_heads, modalities=mods)
        oracle.loss_pairs(out, label).backward()
        opt.step()
        return time.perf_counter() - t0

    run()                                           # warm-up (allocator, page faults), untimed
    times = [run() for _ in range(iters)]
    mean = sum(times) / len(times)
    res = {"value": round(batch / mean, 4), "unit": "tri-modal images/sec", "cores": torch.get_num_threads(),
           "kind": "port",
           "sample": f"oracle fwd + loss_pairs + bwd + SGD step, fp32, B={batch} {nmod}-modal {h}x{w}, 1 warm-up + "
                     f"{iters} timed iterations ({', '.join('%.1f' % x for x in times)} s)"}
    # config 1 (BASELINE.json configs[0]): B=32, RGB only, backbone forward (plumbing check of the reference's CPU path)
    b1 = 32
    x1, _, cam1, _ = synth.make_batch(1111, b1, h, w, cams)
    x1 = x1["RGB"]
    with torch.no_grad():
        oracle.vit_forward(sd, x1[:4], cam1[:4], heads)
        t1 = []
        for _ in range(3):
            t0 = time.perf_counter()
            oracle.vit_forward(sd, x1, cam1, heads)
            t1.append(time.perf_counter() - t0)
    res["c1"] = {"value": round(b1 / (sum(t1) / len(t1)), 2), "unit": "images/sec",
                 "sample": f"config 1: B={b1} RGB-only backbone forward, 3 timed iterations after a warm-up"}
    return res


def _free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def _rank_tails(logdir, nlines=25):
    """the last lines of every rank's stderr.log under torch.distributed.run's --log-dir, rank by rank"""
    import glob
    files = sorted(glob.glob(os.path.join(logdir, "**", "stderr.log"), recursive=True))
    for f in files:
        try:
            tail = open(f, errors="replace").read().splitlines()[-nlines:]
        except OSError:
            continue
        sys.stderr.write(f"---- {os.path.relpath(f, logdir)}\n" + "\n".join(tail) + "\n")
    if not files:
        sys.stderr.write(f"(no rank logs under {logdir})\n")
    sys.stderr.flush()


def launch_ranks(n, argv):
    """`python bench.py --gpus N` without a launcher around it: start the N ranks ourselves, one process per GPU, exactly as the
    reference is started (train_net.py:45-46,63-64: torch.distributed.launch + NCCL) - re-exec this file under
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port <free port>`, pass the
    ranks' stderr through, relay rank 0's JSON line as the LAST line of stdout and return the launcher's exit code.
    N = 1 (`--spawn` / EDITOR_BENCH_SPAWN=1) goes the same way with a real 1-rank RCCL group, so the launcher is exercised on every
    single-GPU box (tests/test_gpu_bench_contract.py)."""
    import subprocess
    have = torch.cuda.device_count()
    if have < n:
        sys.stderr.write(f"[bench] --gpus {n} asked for, but this node shows {have} GPU(s) (torch.cuda.device_count()); not starting\n")
        return 2
    env = dict(os.environ)
    env["EDITOR_BENCH_RANK_CHILD"] = "1"
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, _usable_cores() // max(n, 1))))
    if n == 1:
        env["EDITOR_FORCE_DDP"] = "1"
    child_argv = [a for a in argv if a != "--spawn"]
    import tempfile
    logdir = tempfile.mkdtemp(prefix="editor_bench_ranks_")
    # --tee 2: every rank's stderr goes to the console AND to <logdir>/.../stderr.log, so that a failed run can show each rank's
    # last lines separately; stdout stays a plain pipe (the launcher prefixes teed lines with "[default<rank>]:") (first contact with an 8-GPU node must be diagnosable from the driver's log alone)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), "--log-dir", logdir, "--tee", "2", os.path.abspath(__file__)] + child_argv
    sys.stderr.write("[bench] launching " + " ".join(cmd[1:]) + "\n")
    sys.stderr.flush()
    timeout = float(os.environ.get("EDITOR_BENCH_LAUNCH_TIMEOUT", "3000"))
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True, env=env, cwd=ROOT, start_new_session=True)
    try:
        stdout, _ = proc.communicate(timeout=timeout)
    except subprocess.TimeoutExpired:
        import signal
        os.killpg(proc.pid, signal.SIGKILL)                # the launcher AND its ranks (own session), nothing else
        stdout, _ = proc.communicate()
        sys.stderr.write(f"[bench] the {n}-rank run did not finish within {timeout:.0f} s; killed; per-rank stderr tails:\n")
        _rank_tails(logdir)
        sys.stdout.write(stdout)
        return 124
    import re
    all_lines = [re.sub(r"^\[[a-z]+\d+\]:", "", ln) for ln in stdout.splitlines()]      # (a teed stdout would carry the rank prefix)
    lines = [ln for ln in all_lines if ln.startswith("{") and '"metric"' in ln]
    other = [ln for ln in all_lines if not (ln.startswith("{") and '"metric"' in ln)]
    if other:
        sys.stderr.write("\n".join(other[-40:]) + "\n")  # RCCL banners etc.: not on stdout, the JSON line is the only line there
    if proc.returncode != 0 or not lines:
        sys.stderr.write(f"[bench] the {n}-rank run failed (rc={proc.returncode}, {len(lines)} JSON line(s)); per-rank stderr tails:\n")
        _rank_tails(logdir)
        return proc.returncode or 1
    try:
        j = json.loads(lines[-1])
        assert j["n_gpus"] == n and j.get("rccl_ranks") == n, (j.get("n_gpus"), j.get("rccl_ranks"))
    except Exception as e:
        sys.stderr.write(f"[bench] rank 0's line does not describe an {n}-rank run ({type(e).__name__}: {e})\n")
        return 1
    print(lines[-1], flush=True)
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=None, help="per-GPU batch (BASELINE: 128; SYNTH4L default 32)")
    ap.add_argument("--preset", default="RGBNT201")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f16", "f16x2", "f16x2s", "f32"],
                    help="f16x2: split-precision forward (fp32-class on the half matrix cores); f16x2s: the same only where the token "
                         "selection depends on it (cfg.MODEL.SPLIT_SCOPE = 'selection')")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-iters", type=int, default=3)
    ap.add_argument("--graph", action="store_true", help="time a hipGraph replay of the captured step in THIS process "
                    "(default for 1 GPU, through a child process; opt-in for N > 1, where the eager step is the default)")
    ap.add_argument("--no-graph", action="store_true", help="time the eager step (no hipGraph attempt)")
    ap.add_argument("--no-replay", action="store_true", help="skip the GEMM replay / kernel micro-timings (clean rocprof totals)")
    ap.add_argument("--no-h2d", action="store_true", help="time the bare step instead of the reference loop's feeding (no H2D, no per-step sync)")
    ap.add_argument("--act-light", action="store_true", help="activation-light blocks (cfg.MODEL.ACT_LIGHT): 24 B instead of 36 B "
                    "saved per token-row-element (config 5 at B = 64 per GPU)")
    ap.add_argument("--no-modes", action="store_true", help="skip the per-mode block (speed + accuracy of bf16 / f16 / f16x2 / f32)")
    ap.add_argument("--grad-wire", default="f32", choices=["f32", "bf16"], help="N > 1: dtype of the gradient buckets on the wire "
                    "(bf16: 237.8 instead of 475.7 MB per step over the xGMI ring; the strong-scaling series needs it)")
    ap.add_argument("--no-eval", action="store_true", help="skip the forward-only (do_inference) throughput block")
    ap.add_argument("--no-others", action="store_true", help="skip the other BASELINE configs' child runs (`other_configs`)")
    ap.add_argument("--spawn", action="store_true", help="go through the rank launcher even for --gpus 1 (a real 1-rank RCCL group; "
                    "also EDITOR_BENCH_SPAWN=1)")
    args = ap.parse_args()
    if args.batch is None:
        args.batch = 32 if args.preset == "SYNTH4L" else 128
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and (args.gpus > 1 or args.spawn or os.environ.get("EDITOR_BENCH_SPAWN") == "1"):
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))
    if "WORLD_SIZE" in os.environ and int(os.environ["WORLD_SIZE"]) != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but the launcher started WORLD_SIZE={os.environ['WORLD_SIZE']} ranks")

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    force_ddp = os.environ.get("EDITOR_FORCE_DDP") == "1"        # exercises the RCCL path (incl. capture) on 1 GPU
    # Single GPU, default: the step is timed as a hipGraph replay (one graph launch per step instead of ~1100 kernel
    # launches issued from Python: a slow or busy host stretched the 51 ms step to 80 ms on some boxes).  The capture
    # runs in a CHILD process, because a capture the runtime rejects can crash the process instead of raising; if the
    # child does not deliver its JSON line, this process measures the eager step itself.  The child does the same K
    # timed steps between the same synchronisations - every kernel of the eager step is in the graph.
    if world == 1 and not args.graph and not args.no_graph and not force_ddp:
        import subprocess
        try:
            cp = subprocess.run([sys.executable, os.path.abspath(__file__)] + sys.argv[1:] + ["--graph"],
                                stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=2400)
            lines = [ln for ln in cp.stdout.splitlines() if ln.startswith("{") and '"metric"' in ln]
            if cp.returncode == 0 and lines:
                json.loads(lines[-1])
                sys.stderr.write(cp.stderr[-2000:])
                print(lines[-1], flush=True)
                return
            sys.stderr.write(cp.stderr[-3000:])
            sys.stderr.write(f"[bench] hipGraph child failed (rc={cp.returncode}); timing the eager step\n")
        except Exception as e:                                            # timeout, unparsable output ...
            sys.stderr.write(f"[bench] hipGraph child failed ({type(e).__name__}); timing the eager step\n")
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if torch.cuda.device_count() <= local:
        raise SystemExit(f"rank {rank}: LOCAL_RANK {local} but only {torch.cuda.device_count()} GPU(s) visible")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    use_dist = world > 1 or force_ddp
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("NCCL_DEBUG", "WARN")              # RCCL's own diagnosis of a failed ring on stderr, nothing when it works
        import datetime
        t_init = time.perf_counter()
        try:
            # 120 s: a rank that never arrives (wrong GPU visibility, a port in use, IPC mode) must fail with a message, not sit
            # in the store's default half-hour wait
            dist.init_process_group("nccl", init_method="env://", world_size=world, rank=rank, device_id=dev,
                                    timeout=datetime.timedelta(seconds=float(os.environ.get("EDITOR_BENCH_INIT_TIMEOUT", "120"))))
            probe_t = torch.ones(1, device=dev)
            dist.all_reduce(probe_t)                              # the first collective builds the ring: fail HERE, with context
            torch.cuda.synchronize()
            assert int(probe_t.item()) == world, f"all-reduce of ones over {world} rank(s) gave {probe_t.item()}"
        except Exception as e:
            sys.stderr.write(f"[bench] rank {rank}/{world} (GPU {local}, MASTER {os.environ.get('MASTER_ADDR')}:{os.environ.get('MASTER_PORT')}, "
                             f"HSA_ENABLE_IPC_MODE_LEGACY={os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY')}) could not join the RCCL group "
                             f"after {time.perf_counter() - t_init:.0f} s: {type(e).__name__}: {e}\n")
            sys.stderr.flush()
            raise
        if rank == 0:
            sys.stderr.write(f"[bench] RCCL group of {world} rank(s) up in {time.perf_counter() - t_init:.1f} s\n")

    from editor_amd import config, losses, synth
    from editor_amd.modeling import make_model

    cfg, num_class, cams = config.preset(args.preset, compute_dtype=args.dtype, drop_path=0.1, act_light=args.act_light)
    torch.manual_seed(1111)                                       # SOLVER.SEED (config/defaults.py:138)
    import contextlib
    import io
    with contextlib.redirect_stdout(io.StringIO()):
        model = make_model(cfg, num_class, cams)
    synth.fill_state_dict_(model.state_dict(), 1111)
    model = model.to(dev).train()
    # Multi-GPU: pure data parallelism, one AVG all-reduce of the 118.9 M gradients per step.  The
    # gradients are written by the backward straight into a few flat 64 MiB buckets and each bucket's RCCL all-reduce is
    # issued from inside the backward as soon as its last block is done (editor_amd.ddp.GradBuckets): overlapped with the
    # remaining backward in the eager step AND in the captured hipGraph (the collectives are part of the graph).
    # (N = 1 as well: the in-place gradient slots are how the backward hands its weight gradients over - no process group, no
    # collective; editor_amd.functional.GROUP_WGRAD)
    buckets = model.enable_grad_buckets(force=force_ddp, wire_dtype=torch.bfloat16 if args.grad_wire == "bf16" else None)
    buckets.broadcast_parameters(model)

    # solver/make_optimizer.py:4-29: SGD, momentum 0.9, wd 1e-4, bias lr x2 (BASE_LR 0.001) - fused HIP update
    from editor_amd import solver
    opt, _ = solver.make_optimizer(cfg, model, None)

    h, w = cfg.INPUT.SIZE_TRAIN
    b = args.batch
    nmod = model.nmod
    keys = config.MODALITY_KEYS[:nmod]
    img_h, label, cam, view = synth.make_batch(1111 + rank, b, h, w, cams, instances=min(16, b), keys=keys)
    img = {k: v.to(dev) for k, v in img_h.items()}
    label, cam, view = label.to(dev), cam.to(dev), view.to(dev)
    writer = _Writer()

    def step():
        opt.zero_grad(set_to_none=True)
        if buckets is not None and os.environ.get("EDITOR_BCAST_BUFFERS", "1") != "0":
            buckets.broadcast_buffers(model)       # DDP's per-forward broadcast_buffers (train_net.py:63-64: the default)
        out = model(img, label=label, cam_label=cam, view_label=view, img_path=None, writer=writer, epoch=1)
        loss = losses.loss_pairs(out, label)
        loss.backward()
        if buckets is not None:
            buckets.finish()
        opt.step()
        return loss

    # N > 1 (or EDITOR_FORCE_DDP): the EAGER step is timed unless --graph asks for the captured one.  Measured with a 1-rank RCCL
    # group on one GPU, same box: eager 46.3 ms, graph replay with the bucket all-reduces inside 47.9 ms (the graph executor
    # serialises the comm-stream branch more than the streams themselves do), and at N = 1 eager and replay tie (2 925 vs 2 937
    # img/s) - so the captured form buys nothing there and is the one path no multi-rank box has ever run.
    want_graph = args.graph
    side = torch.cuda.Stream() if want_graph else None
    if want_graph:
        # every eager step before the capture runs on a SIDE stream: AccumulateGrad nodes remember the stream they were
        # created on, and one created on the default stream invalidates a later capture (torch warns; ROCm then crashes
        # in hipStreamEndCapture)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(args.warmup):
                step()
        torch.cuda.current_stream().wait_stream(side)
    else:
        for _ in range(args.warmup):
            step()
    probe = _GemmProbe()
    probe.install()
    # the whole step (forward, loss, backward incl. the bucket all-reduces, fused SGD: ~1100 launches) is captured once
    # into a hipGraph and the timed region replays it; the drop-path generator and the SGD pointer table are replay-safe
    # (device-resident counter, captured upload).
    graph = None
    if want_graph:
        try:
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                step()                                            # allocator / lazy-attribute warm-up on the capture stream
                probe.recording = rank == 0 and not args.no_replay   # the GEMM launch list of one (eager) step: it keeps that
                                                                  # step's operands alive for the replay (--no-replay: config 5 at B = 64)
                step()
                probe.recording = False
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            opt.zero_grad(set_to_none=True)
            # the eager warm-up steps leave their activations CACHED in the allocator's ordinary pool; the capture allocates
            # the same set again from the graph's private pool - hand the cached blocks back first, or large configurations
            # (config 5 at B = 64: ~120 GB of saved activations) need twice their memory
            torch.cuda.empty_cache()
            graph = torch.cuda.CUDAGraph()
            from editor_amd.ddp import graph_capture_kwargs
            with torch.cuda.graph(graph, **graph_capture_kwargs()):        # (RCCL in the step: thread-local capture mode)
                static_loss = step()
            graph.replay()                                        # one untimed replay
            torch.cuda.synchronize()
            if rank == 0:
                print("[bench] timed region = hipGraph replay of the captured step" +
                      (" (RCCL bucket all-reduces inside the graph)" if use_dist else ""), file=sys.stderr)
        except Exception as e:                                    # capture unsupported here: fall back to eager timing
            print(f"[bench] hipGraph capture failed ({type(e).__name__}: {e}); timing the eager step", file=sys.stderr)
            graph = None
            probe.calls = []
            torch.cuda.synchronize()
    if use_dist and want_graph:
        # all ranks must time the SAME path
        okf = torch.tensor([1.0 if graph is not None else 0.0], device=dev)
        dist.all_reduce(okf, op=dist.ReduceOp.MIN)
        if okf.item() < 0.5 and graph is not None:
            graph = None
            probe.calls = []
    # ---- timed region (SURVEY.md 8(d) = the reference's own log line, engine/processor.py:107,114-118): every iteration
    # receives its batch from pinned host memory (:73-78: the image tensors and labels go to the device every iteration) and
    # ends with a device synchronisation; throughput = B / mean seconds per iteration.  The copy is pipelined the way a
    # prefetching loader does it: batch i+1 travels host -> device staging on a copy stream while step i runs, and a
    # device-to-device copy moves it into the step's (static, graph-captured) input tensors.  K copies of 151 MB and K
    # synchronisations are inside the timed region.  --no-h2d times the bare step (inputs resident, one sync at the end).
    feed = not args.no_h2d
    if feed:
        pins = [{k: v.clone().pin_memory() for k, v in img_h.items()} for _ in range(2)]
        lab_pin = [t_.cpu().clone().pin_memory() for t_ in (label, cam, view)]
        stage = [({k: torch.empty_like(v) for k, v in img.items()}, [torch.empty_like(t_) for t_ in (label, cam, view)])
                 for _ in range(2)]
        copy_s = torch.cuda.Stream()

        def prefetch(i):                                  # host -> staging[i & 1] on the copy stream
            with torch.cuda.stream(copy_s):
                for k in img:
                    stage[i & 1][0][k].copy_(pins[i & 1][k], non_blocking=True)
                for dst, src in zip(stage[i & 1][1], lab_pin):
                    dst.copy_(src, non_blocking=True)

        def take(i):                                      # staging[i & 1] -> the step's inputs, on the main stream
            torch.cuda.current_stream().wait_stream(copy_s)
            for k in img:
                img[k].copy_(stage[i & 1][0][k], non_blocking=True)
            for dst, src in zip((label, cam, view), stage[i & 1][1]):
                dst.copy_(src, non_blocking=True)

    if feed:
        # the feed path itself (pinned -> staging on the copy stream, staging -> inputs, replay, device sync) once per staging
        # buffer, untimed: on a fresh box its first pass pages in the copy kernels and the pinned buffers (measured: one
        # 80 - 120 ms stall inside an 8-step timed region when it was first exercised there)
        prefetch(0)
        for i in range(2):
            take(i)
            prefetch(i + 1)
            if graph is not None:
                graph.replay()
            else:
                step()
            torch.cuda.synchronize()

    def timed_loop(feeding):
        import gc
        iter_times = [] if os.environ.get("EDITOR_BENCH_ITER_TIMES") else None
        # a cyclic-GC pass of the interpreter (tens of thousands of autograd / ctypes objects alive) landed in the first timed
        # iteration of one run in three at B = 16 (68.7 ms against 10.7): collect before the region, none inside it
        gc.collect()
        gc.disable()
        try:
            return _timed_loop(feeding, iter_times)
        finally:
            gc.enable()

    def _timed_loop(feeding, iter_times):
        if feeding:
            prefetch(0)
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        loss_ = None
        for i in range(args.steps):
            if feeding:
                take(i)
                prefetch(i + 1)                           # next batch under this step (staging[(i+1)&1] was consumed at i-1)
            if graph is not None:
                graph.replay()
                loss_ = static_loss
            else:
                probe.recording = rank == 0 and i == args.steps - 1 and not probe.calls and not args.no_replay
                loss_ = step()
                probe.recording = False
            if feeding:
                torch.cuda.synchronize()                  # engine/processor.py:107
                if iter_times is not None:
                    iter_times.append(time.perf_counter())
        torch.cuda.synchronize()
        own = time.perf_counter() - t0                    # this rank's K steps, before it waits for the others
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()
        if feeding and iter_times and rank == 0:          # EDITOR_BENCH_ITER_TIMES=1: per-iteration wall times to stderr
            ts = [t0] + iter_times
            print("iteration ms: " + " ".join("%.1f" % (1e3 * (ts[j + 1] - ts[j])) for j in range(len(ts) - 1)), file=sys.stderr)
        el_ = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
        if use_dist:
            dist.all_reduce(el_, op=dist.ReduceOp.MAX)
            owns = [torch.zeros(1, dtype=torch.float64, device=dev) for _ in range(world)]
            dist.all_gather(owns, torch.tensor([own], dtype=torch.float64, device=dev))
            rank_ms[:] = [round(1e3 * float(o.item()) / args.steps, 3) for o in owns]
        return float(el_.item()), loss_

    rank_ms = []
    elapsed, loss = timed_loop(feed)
    replay_only = None
    if feed and world == 1 and not force_ddp:            # the bare step beside it (inputs resident, no per-iteration sync)
        el2, _ = timed_loop(False)
        replay_only = {"value": round(b * args.steps / el2, 2), "ms_per_step": round(1e3 * el2 / args.steps, 3),
                       "what": "same K steps with the inputs resident in HBM and one synchronisation at the end"}
    probe.recording = False
    situ = None
    if rank == 0 and world == 1 and not force_ddp and not args.no_replay and args.dtype != "f32":
        if graph is not None:                             # (eager steps of a capturing process stay off the default stream, see above)
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                situ = probe.in_situ(step)
            torch.cuda.current_stream().wait_stream(side)
        else:
            situ = probe.in_situ(step)                    # one more (untimed) eager step, every GEMM launch timed in place
    probe.remove()
    lossv = float(loss.detach())
    eval_block = None
    if rank == 0 and world == 1 and not force_ddp and not args.no_eval:
        eval_block = eval_throughput(model, img, cam, view, b, max(args.steps, 3))

    if rank == 0:
        kinds = {} if args.no_replay else probe.replay()
        flops = sum(v[0] for v in kinds.values())
        ms = sum(v[1] for v in kinds.values())
        launches = sum(v[2] for v in kinds.values())
        achieved_replay = flops / (ms * 1e-3) / 1e12 if ms > 0 else None  # (--no-replay: not measured, not zero)
        # `achieved` / `frac` = the family INSIDE the step (launch-by-launch HIP events of one eager step, weight gradients on the main
        # stream): what the rocprofv3 summary of the same command shows (profiles/rNN_bench_kernel_stats_serial.csv).
        # The back-to-back replay of the same launches (no other kernel in between: warmer caches) stays beside it as `*_replay`.
        s_exec = None
        if situ:
            s_flops, s_ms = sum(v[0] for v in situ.values()), sum(v[1] for v in situ.values())
            s_exec = sum(v[3] for v in situ.values())
            achieved = s_flops / (s_ms * 1e-3) / 1e12 if s_ms > 0 else None
        else:
            s_flops, s_ms, achieved = flops, ms, achieved_replay
        ms_step = 1e3 * elapsed / args.steps
        arch = cfg.MODEL.TRANSFORMER_TYPE.replace("_patch16_224", "").replace("vit_", "ViT-").replace("base", "B").replace("large", "L")
        # `traffic` / `alg_bytes_per_step` are NOT measured in this run: PMC counters need their own rocprofv3 passes
        # (MI355X_MICROARCH.md), so the figure is read from the newest committed profile of this same command
        # (tools/pmc_traffic.sh -> profiles/rNN_pmc_traffic.json) and labelled as such (`traffic_source`)
        import glob
        traffic, tsrc = None, None
        tfiles = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_pmc_traffic.json")))
        if tfiles and args.preset == "RGBNT201" and b == 128 and args.dtype == "bf16":
            try:
                traffic = json.load(open(tfiles[-1]))
                tsrc = os.path.relpath(tfiles[-1], ROOT) + " (builder-run rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command, tools/pmc_traffic.sh; not measured in this run)"
            except Exception:
                traffic = None
        roof = {"bound": "mfma", "achieved": None if achieved is None else round(achieved, 2), "peak": PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": None if achieved is None else round(achieved / PEAK_TFLOPS, 4),
                "traffic": None if traffic is None else traffic.get("gemm_hbm_bytes_per_step"),
                "traffic_source": tsrc,
                "traffic_note": None if traffic is None else traffic.get("note"),
                "kernel": "16-bit GEMM family: gemm_bf16_pp_kernel (256x256x64 ping-pong; fwd, dgrad, and wgrad as one round of "
                          "split-K workgroups + slab reduction), v_mfma_f32_16x16x32_" + ("bf16" if args.dtype == "bf16" else "f16") +
                          ("; forward products as split-precision half pairs (3 MFMA passes per algorithmic FLOP)" if args.dtype.startswith("f16x2") else ""),
                "achieved_source": "in situ: HIP events around every 16-bit GEMM launch of one eager step" if situ else
                                   "back-to-back replay of the step's recorded GEMM launches",
                "launches_per_step": sum(v[2] for v in situ.values()) if situ else launches,
                "gemm_ms_per_step": round(s_ms, 3), "alg_tflop_per_step": round(s_flops / 1e12, 2),
                # stochastic depth (drop_path 0.1): the reference multiplies the dropped samples' branches by 0 AFTER computing them -
                # `achieved` / `frac` price the reference's algorithmic FLOPs; what the launches really multiplied (the MLP branch runs
                # on the kept samples only, cfg.MODEL.DROP_SKIP) is `executed_*`: the matrix cores' own rate
                "executed_tflop_per_step": None if s_exec is None else round(s_exec / 1e12, 2),
                "achieved_executed": None if (s_exec is None or s_ms <= 0) else round(s_exec / (s_ms * 1e-3) / 1e12, 2),
                "frac_executed": None if (s_exec is None or s_ms <= 0) else round(s_exec / (s_ms * 1e-3) / 1e12 / PEAK_TFLOPS, 4),
                "drop_skip": bool(getattr(model, "drop_skip", False)),
                "achieved_replay": None if achieved_replay is None else round(achieved_replay, 2),
                "frac_replay": None if achieved_replay is None else round(achieved_replay / PEAK_TFLOPS, 4),
                "gemm_ms_per_step_replay": round(ms, 3),
                "by_kind_in_situ": None if not situ else {k: {"tflops": round(f / (m_ * 1e-3) / 1e12, 1), "ms_per_step": round(m_, 3),
                                                              "launches": n, "tflops_executed": round(fx / (m_ * 1e-3) / 1e12, 1)}
                                                          for k, (f, m_, n, fx) in situ.items()},
                "alg_bytes_per_step": None if traffic is None else traffic.get("gemm_alg_bytes_per_step"),
                "by_kind": {k: {"tflops": round(f / (m_ * 1e-3) / 1e12, 1), "ms_per_step": round(m_, 3), "launches": n}
                            for k, (f, m_, n) in kinds.items()}}
        # whole-step view: SURVEY.md 8(d)'s algorithmic FLOPs of one step (240.5 GF per tri-modal 256x128 ViT-B image fwd+bwd,
        # scaled by tokens / width for the other presets through the measured GEMM list when available) over the step time
        alg_step = {"RGBNT201": 240.5e9, "RGBNT100": 240.5e9, "MSVR310": 368e9, "SYNTH4L": 4.4e12}.get(args.preset)
        if alg_step is not None:
            roof["step_alg_tflop"] = round(alg_step * b / 1e12, 2)
            roof["step_frac"] = round(alg_step * b / (ms_step * 1e-3) / 1e12 / PEAK_TFLOPS, 4)
        if args.dtype == "f32":
            roof.update(bound="mfma (exact-f32 parity mode: v_mfma_f32_16x16x4_f32; not the performance path)",
                        achieved=None, frac=None, peak=PEAK_F32_TFLOPS)
        if graph is not None and roof.get("frac") is not None:
            # the peak above is priced at the 2.4 GHz boost clock; under this step's load the chip clocks to its power budget
            try:
                mhz = shader_clock_mhz(graph.replay)
            except Exception:
                mhz = None
            if mhz:
                roof["sclk_mhz_during_step"] = round(mhz[0], 0)
                roof["sclk_mhz_min_max_0p5ms"] = None if mhz[1] is None else [round(mhz[1][0], 0), round(mhz[1][1], 0)]
        if not args.no_replay:
            roof["hbm_kernels"], roof["hbm_kernels_in_situ_source"] = hbm_kernels(model, img, b, model.act_dtype, probe.entries)
            if probe.entries:               # the step's non-GEMM time by library entry, this run (ms per step, calls per step)
                other = {k: v for k, v in probe.entries.items() if not k.startswith("editor_gemm")}
                roof["non_gemm_in_situ"] = {"ms_per_step": round(sum(v[0] for v in other.values()), 3),
                                            "top": {k: [round(v[0], 3), round(v[1], 1)] for k, v in
                                                    sorted(other.items(), key=lambda kv: -kv[1][0])[:(200 if os.environ.get("EDITOR_BENCH_ENTRIES") else 14)]}}
            roof["hbm_kernels_note"] = ("us / frac: this run, HIP events, every kernel rotating over `sets` operand sets > 512 MB (HBM, not the "
                                        "256 MB Infinity Cache); in_situ_*: the same kernel's average duration inside the step (source: "
                                        "hbm_kernels_in_situ_source) - there its operands were written by the previous kernel and are partly "
                                        "cache-resident (a fraction above 1 is that, not a faster HBM)")
        out = {
            "metric": "tri-modal images/sec fwd+bwd @ B=128 ViT-B",
            "value": round(world * b * args.steps / elapsed, 2),
            "value_definition": ("B / mean seconds per iteration of the reference's loop (engine/processor.py:73-78,107,114-118): "
                                 "H2D of the batch every iteration (pipelined) + the step + a device sync per iteration")
            if feed else "bare step: inputs resident, one synchronisation after K steps",
            "unit": "tri-modal images/sec" if nmod == 3 else f"{nmod}-modal images/sec",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_step, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"{args.preset} {nmod}-modal {arch}/16 {h}x{w}, batch {b}/GPU, fwd+bwd+SGD step, "
                                   f"drop_path 0.1, SFTS+HMA HIP kernels" + (", activation-light blocks" if args.act_light else ""),
                       "global_batch": world * b, "parallelism": f"dp{world}", "loss": round(lossv, 4), "graph": graph is not None,
                       "launch": ("hipGraph replay" + (" incl. RCCL bucket all-reduces (overlapped with backward)" if use_dist else ""))
                       if graph is not None else ("eager" + (", RCCL bucket all-reduces overlapped with backward" if use_dist else "")),
                       "grad_buckets": None if buckets is None else buckets.describe()},
            "roofline": roof,
        }
        if use_dist:
            # what the driver needs to see that RCCL really ran N ranks: the group's size as torch.distributed reports it after
            # init, and every rank's own time for its K steps (before the closing barrier; `ms_per_step` is the max incl. it)
            out["rccl_ranks"] = dist.get_world_size()
            out["rank_ms_per_step"] = {"min": min(rank_ms), "max": max(rank_ms), "all": rank_ms}
            out["not_in_this_line"] = ("modes, cpu_baseline, replay_only and torch_gpu_yardstick are single-process blocks of the "
                                       "N = 1 line (python bench.py --gpus 1)")
        if eval_block is not None:
            out["eval"] = eval_block
        if replay_only is not None:
            out["replay_only"] = replay_only
        if world == 1 and not args.no_modes and not force_ddp and args.preset in ("RGBNT201", "RGBNT100", "MSVR310"):
            out["modes"] = modes_block(args, cfg, cams, out["value"], eval_block)
            out["value_at_parity"] = out["modes"].pop("value_at_parity")
        if (world == 1 and not force_ddp and not args.no_others and not args.no_modes and args.preset == "RGBNT201" and b == 128
                and args.dtype != "f32"):
            out["other_configs"] = other_configs_block(args)       # (with the full default line only: the modes' children pass --no-modes)
        if world == 1 and not args.no_cpu_baseline and not force_ddp:
            out["cpu_baseline"] = cpu_baseline(model, cfg, cams, b, args.cpu_iters)
            if args.preset == "RGBNT201" and b == 128:
                out["torch_gpu_yardstick"] = torch_gpu_yardstick()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        # RCCL writes its version banner through C stdio (block-buffered on a pipe): push it out first, so that the
        # JSON line is the LAST line on stdout
        import ctypes
        ctypes.CDLL(None).fflush(None)
        sys.stdout.flush()
        print(json.dumps(out), flush=True)
    if use_dist:
        # RCCL's banner sits in a C++ stream buffer that is only flushed by the static destructors at interpreter exit,
        # i.e. AFTER the JSON line: leave without running them (everything of ours is flushed and the group is destroyed)
        sys.stdout.flush()
        sys.stderr.flush()
        os._exit(0)


if __name__ == "__main__":
    main()
