"""Builds the HIP libraries of the hot path for gfx950 with hipcc.

    editor_amd/libeditor_hip.so          every kernel of the product path (csrc/*.hip except probe.hip)
    editor_amd/libeditor_probe.so        csrc/probe.hip: hardware-semantics probes (tests only, include/editor_debug.h)
    editor_amd/libeditor_gemm_trace.so   csrc/gemm_bf16.hip with -DEDITOR_DEBUG_TRACE (tools/gemm_bench.py; --trace only)

In-tree, explicit hipcc (no JIT cache): the .so files travel to the GPU box with the repo snapshot.
    python -m editor_amd.build [--force] [--trace]
"""
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libeditor_hip.so")
LIB_PROBE = os.path.join(HERE, "libeditor_probe.so")
LIB_TRACE = os.path.join(HERE, "libeditor_gemm_trace.so")
LIB_ALT = os.path.join(HERE, "libeditor_gemm_alt.so")      # csrc/gemm_bf16.hip with the OTHER K-tile choreography (A/B runs, tools/gemm_bench.py)
ALT_FLAGS = ["-DEDITOR_PP_PHASES=%s" % os.environ.get("EDITOR_ALT_PHASES", "4")]
LIB_MI32 = os.path.join(HERE, "libeditor_gemm_mi32.so")    # ... with v_mfma_f32_32x32x16 in the full-tile forward / dgrad products (round 6 A/B)
MI32_FLAGS = ["-DEDITOR_PP_MI32=1"]
LIB_ATTN_ALT = os.path.join(HERE, "libeditor_attn_alt.so")     # csrc/attention_bf16.hip with both padding-tile switches flipped (tools/attn_ab.py)
LIB_MI32P2 = os.path.join(HERE, "libeditor_gemm_mi32p2.so")  # ... and the two-phase K-tile (16-MFMA clusters over four accumulators)
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function",
         "-munsafe-fp-atomics"]
DEBUG_ONLY = {"probe.hip", "gemm_w4.hip"}          # libeditor_probe.so: tests / tools only


def _ab_sources(lib):
    stem = "attention_bf16.hip" if "attn" in os.path.basename(lib) else "gemm_bf16.hip"
    return [os.path.join(CSRC, stem), os.path.join(CSRC, "common.h"), os.path.join(CSRC, "attn_common.h"),
            os.path.join(HERE, "..", "include", "editor_hip.h")]


def ab_source_hash(lib):
    """sha1 of the sources an optional A/B build (--alt, --mi32: gemm_bf16.hip; --attn-alt: attention_bf16.hip) is made of.  Written
    beside it as <lib>.srchash at link time: the tests / tools that compare such a build with the product library skip when it does not
    match the CURRENT sources (file times do not survive the copy to the GPU box)."""
    import hashlib
    hsh = hashlib.sha1()
    for f in _ab_sources(lib):
        hsh.update(open(f, "rb").read())
    return hsh.hexdigest()[:16]


def ab_lib_current(lib):
    stamp = lib + ".srchash"
    return os.path.exists(lib) and os.path.exists(stamp) and open(stamp).read().strip() == ab_source_hash(lib)


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _compile(jobs, verbose):
    procs = []
    for src, obj, extra in jobs:
        cmd = [HIPCC] + FLAGS + extra + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd))
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            sys.stderr.write(out.decode())
            raise RuntimeError("hipcc failed on " + src)
        if verbose and out:
            print(out.decode())


def _link(lib, objs):
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs)


def build(force=False, verbose=False, trace=False, alt=False, mi32=False, attn_alt=False):
    hdrs = glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(HERE, "..", "include", "*.h"))
    srcs = sorted(s for s in glob.glob(os.path.join(CSRC, "*.hip")) if os.path.basename(s) not in DEBUG_ONLY)
    objs = [s[:-4] + ".o" for s in srcs]
    jobs = [(s, o, []) for s, o in zip(srcs, objs) if force or _stale(o, [s] + hdrs)]
    # the fused 16-bit and split-precision attention kernels once more per extra head width (csrc/attn_common.h: ATTN_HD; the plain object is the
    # 64-wide build with the C-ABI names, which forwards hd = 32 / 96 to these)
    for stem in ("attention_bf16", "attention_split"):
        attn_src = os.path.join(CSRC, stem + ".hip")
        for hd in (32, 96):
            obj = os.path.join(CSRC, "%s.hd%d.o" % (stem, hd))
            objs.append(obj)
            if force or _stale(obj, [attn_src] + hdrs):
                jobs.append((attn_src, obj, ["-DATTN_HD=%d" % hd]))
    probe_objs = []
    for name in sorted(DEBUG_ONLY):
        probe_src = os.path.join(CSRC, name)
        probe_obj = probe_src[:-4] + ".o"
        probe_objs.append(probe_obj)
        if force or _stale(probe_obj, [probe_src] + hdrs):
            jobs.append((probe_src, probe_obj, []))
    trace_obj = os.path.join(CSRC, "gemm_bf16.trace.o")
    gemm_src = os.path.join(CSRC, "gemm_bf16.hip")
    if trace and (force or _stale(trace_obj, [gemm_src] + hdrs)):
        jobs.append((gemm_src, trace_obj, ["-DEDITOR_DEBUG_TRACE"]))
    alt_obj = os.path.join(CSRC, "gemm_bf16.alt.o")
    if alt and (force or _stale(alt_obj, [gemm_src] + hdrs)):
        jobs.append((gemm_src, alt_obj, ALT_FLAGS))
    attnalt_objs = []
    if attn_alt:
        attn_src = os.path.join(CSRC, "attention_bf16.hip")
        for hd, flags in ((64, []), (32, ["-DATTN_HD=32"]), (96, ["-DATTN_HD=96"])):
            obj = os.path.join(CSRC, "attention_bf16.alt%d.o" % hd)
            attnalt_objs.append(obj)
            if force or _stale(obj, [attn_src] + hdrs):
                jobs.append((attn_src, obj, flags + ["-DATTN_PAIR_SKIP=1", "-DATTN_ROLLOUT_SKIP=0"]))
    mi32_obj = os.path.join(CSRC, "gemm_bf16.mi32.o")
    if mi32 and (force or _stale(mi32_obj, [gemm_src] + hdrs)):
        jobs.append((gemm_src, mi32_obj, MI32_FLAGS))
    mi32p2_obj = os.path.join(CSRC, "gemm_bf16.mi32p2.o")
    if mi32 and (force or _stale(mi32p2_obj, [gemm_src] + hdrs)):
        jobs.append((gemm_src, mi32p2_obj, MI32_FLAGS + ["-DEDITOR_PP_MI32_PHASES=2"]))
    _compile(jobs, verbose)
    if force or _stale(LIB, objs):
        _link(LIB, objs)
    if force or _stale(LIB_PROBE, probe_objs):
        _link(LIB_PROBE, probe_objs)
    if trace and (force or _stale(LIB_TRACE, [trace_obj])):
        _link(LIB_TRACE, [trace_obj])
    if alt and (force or _stale(LIB_ALT, [alt_obj])):
        _link(LIB_ALT, [alt_obj])
        open(LIB_ALT + ".srchash", "w").write(ab_source_hash(LIB_ALT) + "\n")
    if attn_alt and (force or _stale(LIB_ATTN_ALT, attnalt_objs)):
        _link(LIB_ATTN_ALT, attnalt_objs)
        open(LIB_ATTN_ALT + ".srchash", "w").write(ab_source_hash(LIB_ATTN_ALT) + "\n")
    if mi32 and (force or _stale(LIB_MI32, [mi32_obj])):
        _link(LIB_MI32, [mi32_obj])
        open(LIB_MI32 + ".srchash", "w").write(ab_source_hash(LIB_MI32) + "\n")
    if mi32 and (force or _stale(LIB_MI32P2, [mi32p2_obj])):
        _link(LIB_MI32P2, [mi32p2_obj])
        open(LIB_MI32P2 + ".srchash", "w").write(ab_source_hash(LIB_MI32P2) + "\n")
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True, trace="--trace" in sys.argv, alt="--alt" in sys.argv, mi32="--mi32" in sys.argv, attn_alt="--attn-alt" in sys.argv))
