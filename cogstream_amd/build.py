"""Build libcogs_hip.so (gfx950) in-tree with hipcc. No torch involved: the library is plain HIP + C ABI.

    python -m cogstream_amd.build          # incremental
    python -m cogstream_amd.build --force  # rebuild everything
"""
from __future__ import annotations

import os
import re
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

HERE = Path(__file__).resolve().parent
CSRC = HERE / "csrc"
OBJ = HERE / "csrc" / "build"
LIB = HERE / "libcogs_hip.so"
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-Wall", "-Wno-unused-function"]
# per-source additions. attn_vit.hip: its row maxima run on MFMA results, which hipcc otherwise canonicalises (v_max x, x)
# in front of every v_max3 chain -- four extra vector instructions per 32-key block on the port that bounds the kernel;
# scores are finite or -inf (masked), never NaN, and the kernel's own NaN-sensitive test (`!(d > -inf)`) keeps its
# meaning for -inf. Consequence for callers: a NaN in q / k / v is NOT reliably propagated by this kernel (its maxima may drop
# it); tests that trace out-of-range reads therefore use huge finite sentinels (tests/test_gpu_ops.py, head-major guards).
EXTRA_FLAGS = {"attn_vit": ["-fno-honor-nans"]}


def _stale(target: Path, deps) -> bool:
    if not target.exists():
        return True
    t = target.stat().st_mtime
    return any(d.stat().st_mtime > t for d in deps)


OBJDUMP = os.environ.get("LLVM_OBJDUMP", "/opt/rocm/lib/llvm/bin/llvm-objdump")
EPI_BIAS, EPI_RES, EPI_ROPE, EPI_SWIGLU, EPI_ROPE_LUT, EPI_ROWSTAT, EPI_LNFOLD = 1, 2, 4, 32, 512, 1024, 2048      # csrc/gemm_epilogue.h
_VMEM = re.compile(r"^\s*(global_load|global_store|buffer_load|buffer_store|flat_load|flat_store|scratch_)")


def epi_pair_vmem_ops(epi: int) -> int:
    """csrc/gemm_epilogue.h::epi_pair_vmem_ops<EPI>()"""
    if epi & EPI_SWIGLU:
        return 8
    return 16 + (4 if epi & EPI_BIAS else 0) + (16 if epi & EPI_RES else 0) + \
        ((8 if epi & EPI_ROPE_LUT else 32) if epi & EPI_ROPE else 0) + (8 if epi & EPI_LNFOLD else 0) + \
        (8 if epi & EPI_ROWSTAT else 0)


def check_epilogue_vmem_counts(obj: Path):
    """The ping-pong GEMM relaxes the first s_waitcnt vmcnt of a tile by the number of vector-memory instructions the
    previous tile's epilogue issued (csrc/gemm.hip, wait_next_ktile). That number is a compile-time formula; the
    epilogue's loads and stores are plain C++, so this check disassembles gemm.o and counts what hipcc really
    emitted between the region markers (s_nop 8|9 ... s_nop 10) of every gemm_tn_pp_kernel<EPI>. vmcnt(N) waits until
    at most N operations are outstanding, so the wait stays CORRECT as long as the epilogue issues AT LEAST the
    assumed number (more = the wait also covers the oldest extra ones, e.g. a spill reload at the region start:
    scratch_* instructions count on vmcnt too); fewer than assumed would let LDS-DMA pieces of the next K-tile be
    read before they land. Returns (unsafe, slack, regions): unsafe = [(kernel EPI, region kind, emitted, assumed)]
    with emitted < assumed, slack = the same tuples with emitted > assumed."""
    dis_text = _disassemble(obj, "epi_check")
    bad, slack, regions, kern_epi = [], [], 0, None
    state, count = None, 0
    for line in dis_text.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.*)>:", line)
        if m:
            k = re.search(r"gemm_tn_pp(?:64)?_kernelILi(\d+)E", m.group(1))
            kern_epi, state = (int(k.group(1)) if k else None), None
            continue
        if kern_epi is None:
            continue
        ins = line.split("//")[0].strip()
        if re.match(r"^s_nop (8|9)\b", ins):
            state, count = ("rope" if ins.split()[1] == "9" else "plain"), 0
        elif re.match(r"^s_nop 10\b", ins) and state:
            epi = kern_epi if (state == "rope" or not kern_epi & EPI_ROPE) else kern_epi & ~(EPI_ROPE | EPI_ROPE_LUT)
            want = epi_pair_vmem_ops(epi)
            regions += 1
            if count < want:
                bad.append((kern_epi, state, count, want))
            elif count > want:
                slack.append((kern_epi, state, count, want))
            state = None
        elif state and _VMEM.match(ins):
            count += 1
    if regions == 0:
        raise RuntimeError("no epilogue region markers found in gemm.o (was the marker asm removed?)")
    return bad, slack, regions


def _disassemble(obj: Path, tag: str) -> str:
    """text disassembly of the gfx950 code object bundled in a hipcc object file"""
    tmp = obj.parent / tag
    tmp.mkdir(exist_ok=True)
    work = tmp / obj.name
    work.write_bytes(obj.read_bytes())
    try:
        r = subprocess.run([OBJDUMP, "--offloading", str(work)], capture_output=True, text=True)
        bundles = sorted(tmp.glob(obj.name + ".*gfx950*"))
        if r.returncode != 0 or not bundles:
            raise RuntimeError(f"cannot extract the gfx950 code object of {obj.name}: {r.stdout}{r.stderr}")
        dis = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", str(bundles[0])], capture_output=True, text=True)
        if dis.returncode != 0:
            raise RuntimeError(dis.stderr)
        return dis.stdout
    finally:
        for f in tmp.glob("*"):
            f.unlink()


def check_m0_uses(obj: Path, kernel="attn_vit_pipe_kernel"):
    """csrc/attn_vit.hip issues its LDS-DMA pieces from inline asm that writes M0 (`s_mov_b32 m0, sN` right in front of
    `global_load_lds_dwordx4`) without naming M0 as clobbered (hipcc warns about the clobber and never keeps a value in
    M0 across statements). That assumption is checked here instead: inside attn_vit_pipe_kernel every instruction that
    mentions m0 must be one of those `s_mov_b32 m0, s..` writes -- a compiler-generated read or write of M0 anywhere in
    the kernel would mean the two could interleave. Returns the offending lines."""
    bad, inside = [], False
    for line in _disassemble(obj, "m0_check").splitlines():
        m = re.match(r"^[0-9a-f]+ <(.*)>:", line)
        if m:
            inside = any(k in m.group(1) for k in ((kernel,) if isinstance(kernel, str) else kernel))
            continue
        ins = line.split("//")[0].strip()
        if inside and re.search(r"\bm0\b", ins) and not re.match(r"^s_mov_b32 m0, s\d+$", ins):
            bad.append(ins)
    return bad


def build(force: bool = False, verbose: bool = False) -> Path:
    srcs = sorted(CSRC.glob("*.hip"))
    hdrs = sorted(CSRC.glob("*.h")) + [HERE.parent / "include" / "cogs.h"]
    OBJ.mkdir(parents=True, exist_ok=True)
    jobs = []
    for s in srcs:
        o = OBJ / (s.stem + ".o")
        if force or _stale(o, [s] + hdrs):
            jobs.append((s, o))

    def cc(job):
        s, o = job
        cmd = [HIPCC] + FLAGS + EXTRA_FLAGS.get(s.stem, []) + ["-c", str(s), "-o", str(o)]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed on {s.name}:\n{r.stdout}\n{r.stderr}")
        return r.stderr

    if jobs:
        with ThreadPoolExecutor(max_workers=min(4, len(jobs))) as ex:
            for warn in ex.map(cc, jobs):
                if warn and verbose:
                    print(warn)
    strict = os.environ.get("COGS_STRICT_BUILD") == "1"
    gemm_o, stamp = OBJ / "gemm.o", OBJ / "gemm.epi_check.txt"
    # keyed on the stamp being older than the object, not on "compiled in this invocation": a gemm.o left by an
    # interrupted build (or by tools/build_alt.sh) is checked before it is linked
    if "COGS_EPI_NOPAIR" not in " ".join(FLAGS) and (not stamp.exists() or stamp.stat().st_mtime < gemm_o.stat().st_mtime):
        try:
            bad, slack, regions = check_epilogue_vmem_counts(gemm_o)
        except (OSError, RuntimeError) as e:      # llvm-objdump missing / extraction failed: cannot verify -> safe build
            if strict:
                raise
            bad, slack, regions = [(-1, f"check not possible ({e})", 0, 0)], [], 0
        if bad:
            msg = "; ".join(f"kernel<EPI={k}> {kind} region: {got} vector-memory instructions, the wait assumes {want}"
                            for k, kind, got, want in bad)
            if strict:
                raise RuntimeError("epilogue vmem-count check failed: " + msg)
            print("WARNING: epilogue vmem-count check failed (" + msg + "); rebuilding gemm.hip with the conservative "
                  "s_waitcnt (-DCOGS_EPI_CONSERVATIVE)", flush=True)
            cmd = [HIPCC] + FLAGS + ["-DCOGS_EPI_CONSERVATIVE", "-c", str(CSRC / "gemm.hip"), "-o", str(gemm_o)]
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode != 0:
                raise RuntimeError(f"hipcc failed on gemm.hip:\n{r.stdout}\n{r.stderr}")
            stamp.write_text("conservative: " + msg + "\n")
        else:
            extra = "; ".join(f"<EPI={k}> {kind}: {got} emitted vs {want} assumed" for k, kind, got, want in slack)
            stamp.write_text(f"ok: {regions} epilogue regions, every region emits at least the assumed vector-memory "
                             f"instructions ({len(slack)} with more: {extra or 'none'})\n")
            if verbose:
                print(stamp.read_text().strip(), flush=True)
    attn_o, m0_stamp = OBJ / "attn_vit.o", OBJ / "attn_vit.m0_check.txt"
    if not m0_stamp.exists() or m0_stamp.stat().st_mtime < attn_o.stat().st_mtime:
        try:
            m0_bad = check_m0_uses(attn_o)
        except (OSError, RuntimeError) as e:
            if strict:
                raise
            print(f"WARNING: M0 check of attn_vit.o not possible ({e})", flush=True)
            m0_bad = None
        if m0_bad:
            raise RuntimeError("attn_vit_pipe_kernel: hipcc generated its own uses of M0 beside the inline-asm LDS-DMA "
                               "(csrc/attn_vit.hip, dma16): " + "; ".join(m0_bad[:6]))
        if m0_bad is not None:
            m0_stamp.write_text("ok: every m0 reference inside attn_vit_pipe_kernel is an inline-asm s_mov_b32 m0, sN\n")
    # the same inline-asm LDS-DMA in the Qwen2 prompt attention (csrc/attn.hip, attn_prefill_dma_kernel)
    attn2_o, m0_stamp2 = OBJ / "attn.o", OBJ / "attn.m0_check.txt"
    if not m0_stamp2.exists() or m0_stamp2.stat().st_mtime < attn2_o.stat().st_mtime:
        try:
            m0_bad = check_m0_uses(attn2_o, "attn_prefill_dma_kernel")
        except (OSError, RuntimeError) as e:
            if strict:
                raise
            print(f"WARNING: M0 check of attn.o not possible ({e})", flush=True)
            m0_bad = None
        if m0_bad:
            raise RuntimeError("attn_prefill_dma_kernel: hipcc generated its own uses of M0 beside the inline-asm LDS-DMA "
                               "(csrc/attn.hip, dma16): " + "; ".join(m0_bad[:6]))
        if m0_bad is not None:
            m0_stamp2.write_text("ok: every m0 reference inside attn_prefill_dma_kernel is an inline-asm s_mov_b32 m0, sN\n")
    m0_stamp3 = OBJ / "gemm.m0_check.txt"
    if not m0_stamp3.exists() or m0_stamp3.stat().st_mtime < gemm_o.stat().st_mtime:
        try:
            m0_bad = check_m0_uses(gemm_o, "gemm_tn_pp64_kernel")
        except (OSError, RuntimeError) as e:
            if strict:
                raise
            print(f"WARNING: M0 check of gemm.o not possible ({e})", flush=True)
            m0_bad = None
        if m0_bad:
            raise RuntimeError("gemm_tn_pp64_kernel: hipcc generated its own uses of M0 beside the inline-asm LDS-DMA "
                               "(csrc/gemm.hip, issue4): " + "; ".join(m0_bad[:6]))
        if m0_bad is not None:
            m0_stamp3.write_text("ok: every m0 reference inside gemm_tn_pp64_kernel is an inline-asm s_mov_b32 m0, sN\n")
    objs = [OBJ / (s.stem + ".o") for s in srcs]
    if force or jobs or _stale(LIB, objs):
        cmd = [HIPCC, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", str(LIB)] + [str(o) for o in objs]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    return LIB


LIB_NOPAIR = HERE / "libcogs_hip_nopair.so"


def build_nopair(force: bool = False) -> Path:
    """A/B library for tests/test_gpu_ops.py: the same sources with -DCOGS_EPI_NOPAIR, i.e. the ping-pong GEMM stores
    through the generic per-half epilogue and always takes the conservative s_waitcnt. Outputs must equal the default
    build's bit for bit (a wrong relaxed vmcnt would show up as a difference). Only gemm.hip is compiled again."""
    build()
    o = OBJ / "gemm_nopair.o"
    hdrs = sorted(CSRC.glob("*.h")) + [HERE.parent / "include" / "cogs.h"]
    if force or _stale(o, [CSRC / "gemm.hip"] + hdrs):
        cmd = [HIPCC] + FLAGS + ["-DCOGS_EPI_NOPAIR", "-c", str(CSRC / "gemm.hip"), "-o", str(o)]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed on gemm.hip (nopair):\n{r.stdout}\n{r.stderr}")
    objs = [o if s.stem == "gemm" else OBJ / (s.stem + ".o") for s in sorted(CSRC.glob("*.hip"))]
    if force or _stale(LIB_NOPAIR, objs):
        cmd = [HIPCC, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", str(LIB_NOPAIR)] + [str(x) for x in objs]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    return LIB_NOPAIR


if __name__ == "__main__":
    p = build(force="--force" in sys.argv, verbose="-v" in sys.argv)
    print(p)
    if "--nopair" in sys.argv:
        print(build_nopair(force="--force" in sys.argv))
