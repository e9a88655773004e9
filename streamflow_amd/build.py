"""Build libstreamflow_hip.so (HIP kernels + C ABI) for gfx950 with hipcc, in-tree.

    python -m streamflow_amd.build [--force | --clean] [--resources] [--asan]

``--clean`` removes every object, resource record and the library first and compiles all sources from nothing (what a fresh
checkout does; the incremental mode compares mtimes only).

hipcc cross-compiles without a GPU.  Objects go to ``streamflow_amd/csrc/build/`` and the shared
library to ``streamflow_amd/libstreamflow_hip.so`` (git-ignored, shipped to the GPU box by gpurun).
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "build")
LIB = os.path.join(HERE, "libstreamflow_hip.so")
ASAN_OBJ = os.path.join(CSRC, "build_asan")
ASAN_LIB = os.path.join(HERE, "libstreamflow_hip_asan.so")
SOURCES = ["misc.hip", "corr.hip", "corr_blocked.hip", "corr_blocked32.hip", "conv.hip", "gemm.hip", "gemm_split.hip", "gemm_bstat.hip", "ffn_pair.hip", "sk_tail.hip", "temporal.hip", "mask_upsample.hip", "attn.hip", "encoder.hip"]
HEADERS = [os.path.join(CSRC, "sf_common.h"), os.path.join(CSRC, "gemm_epilogue.h"), os.path.join(CSRC, "split_operand.h"),
           os.path.join(HERE, "..", "include", "streamflow_hip.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-slp-vectorize", "-Wall",
         "-Wno-unused-function",
         # per-kernel registers / scratch / occupancy as compiler remarks: kept next to every object (<name>.res) and checked
         # by check_resources() -- a hot kernel that spills must fail the build, not ship
         "-Rpass-analysis=kernel-resource-usage"]
# kernels that may not use scratch memory (substring of the demangled name): everything that shows up in the top rows of the
# step's kernel table (profiles/r*_kernel_stats_sintel_serial.md)
HOT_KERNELS = ("gemm_bstat", "ffn_pair_kernel", "sk_tail_kernel", "gma_flash_pipe_kernel", "gma_pv_kernel", "temporal_block_kernel", "mask_upsample_kernel", "gemm_bdirect_kernel", "gma_flash_kernel", "flash_project_v_kernel", "dwconv_mfma_kernel",
               "corr_lookup_blocked_kernel", "corr_build_blocked_kernel", "corr_lookup_blocked32_kernel", "corr_build_blocked32_kernel", "temporal_attn_kernel", "layernorm_cm_split_kernel",
               "flash_pack_v_kernel")


def hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found; libstreamflow_hip.so cannot be built")
    return exe


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _strip_remarks(stderr: str) -> str:
    """Compiler output without the resource-usage remarks (each is a remark line + two source-context lines)."""
    out, skip = [], 0
    for ln in stderr.splitlines():
        if "kernel-resource-usage" in ln:
            skip = 2
            continue
        if skip and (ln.lstrip().startswith("|") or (ln.lstrip()[:1].isdigit() and "|" in ln)):
            skip -= 1
            continue
        skip = 0
        if "remarks generated" in ln or "remark generated" in ln:
            continue
        out.append(ln)
    return "\n".join(out)


def resources() -> dict:
    """{demangled kernel name: {"sgpr", "vgpr", "agpr", "scratch", "occupancy", "lds", "file"}} from the .res files the
    last build() left next to the objects."""
    import re
    res, cur = {}, None
    keys = {"TotalSGPRs": "sgpr", "VGPRs": "vgpr", "AGPRs": "agpr", "ScratchSize [bytes/lane]": "scratch",
            "Occupancy [waves/SIMD]": "occupancy", "LDS Size [bytes/block]": "lds", "VGPRs Spill": "vgpr_spill",
            "SGPRs Spill": "sgpr_spill"}
    mangled = []
    for src in SOURCES:
        path = os.path.join(OBJ, src.replace(".hip", ".res"))
        if not os.path.exists(path):
            raise RuntimeError(f"{path} missing: run streamflow_amd.build.build(force=True)")
        for ln in open(path):
            m = re.search(r"remark:\s+(.*?):\s+(\S+) \[-Rpass", ln)
            if not m:
                continue
            k, v = m.group(1).strip(), m.group(2)
            if k == "Function Name":
                cur = {"file": src}
                res[v] = cur
                mangled.append(v)
            elif cur is not None and k in keys:
                cur[keys[k]] = int(v)
    filt = shutil.which("llvm-cxxfilt") or shutil.which("c++filt") or "/usr/bin/c++filt"
    if os.path.exists(filt) and mangled:
        names = subprocess.run([filt], input="\n".join(mangled), capture_output=True, text=True).stdout.splitlines()
        if len(names) == len(mangled):
            res = {n.replace("(anonymous namespace)::", ""): res[m_] for n, m_ in zip(names, mangled)}
    return res


def check_resources(hot=HOT_KERNELS) -> dict:
    """Fail when a hot kernel uses scratch memory (register spills or a dynamically indexed register array)."""
    res = resources()
    bad = {n: r for n, r in res.items() if r.get("scratch", 0) > 0 and any(h in n for h in hot)}
    if bad:
        raise RuntimeError("hot kernels with scratch memory (spills): " +
                           "; ".join(f"{n}: {r['scratch']} B/lane, {r.get('vgpr_spill', 0)} VGPRs spilled" for n, r in bad.items()))
    return res


def resources_markdown() -> str:
    res = resources()
    rows = ["| kernel | file | VGPRs | AGPRs | SGPRs | scratch B/lane | waves/SIMD | LDS B/WG |", "|---|---|---:|---:|---:|---:|---:|---:|"]
    for n in sorted(res, key=lambda n: (res[n]["file"], n)):
        r = res[n]
        short = n.split("(")[0] if "<" not in n else n[: n.rfind(">") + 1]
        rows.append(f"| `{short}` | {r['file']} | {r.get('vgpr')} | {r.get('agpr')} | {r.get('sgpr')} | {r.get('scratch')} | "
                    f"{r.get('occupancy')} | {r.get('lds')} |")
    return "\n".join(rows) + "\n"


def clean(asan: bool = False) -> None:
    """Remove the objects, resource records and the shared library (the next build() starts from the sources alone)."""
    for d, lib in ((OBJ, LIB),) + (((ASAN_OBJ, ASAN_LIB),) if asan else ()):
        shutil.rmtree(d, ignore_errors=True)
        if os.path.exists(lib):
            os.remove(lib)


def compile_source(src: str, obj_dir: str, verbose: bool = False) -> str:
    """One translation unit from scratch into `obj_dir` (object + .res record) with the flags of the real build: the unit of
    build(); tests/test_abi_cpu.py compiles the smallest source this way into an empty directory."""
    os.makedirs(obj_dir, exist_ok=True)
    s, o = os.path.join(CSRC, src), os.path.join(obj_dir, src.replace(".hip", ".o"))
    cmd = [hipcc()] + FLAGS + ["-c", s, "-o", o]
    if verbose:
        print("[build]", " ".join(cmd), flush=True)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {s}:\n{r.stdout}\n{r.stderr}")
    with open(o[:-2] + ".res", "w") as f:
        f.write("\n".join(ln for ln in r.stderr.splitlines() if "kernel-resource-usage" in ln) + "\n")
    other = _strip_remarks(r.stderr)
    if verbose and other.strip():
        print(other, file=sys.stderr)
    return o


def build(force: bool = False, verbose: bool = True) -> str:
    os.makedirs(OBJ, exist_ok=True)
    cc = hipcc()
    jobs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJ, src.replace(".hip", ".o"))
        if force or _stale(o, [s] + HEADERS) or not os.path.exists(o[:-2] + ".res"):
            jobs.append((s, o))

    def compile_one(job):
        return compile_source(os.path.basename(job[0]), OBJ, verbose)

    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(compile_one, jobs))
    objs = [os.path.join(OBJ, s.replace(".hip", ".o")) for s in SOURCES]
    if force or jobs or _stale(LIB, objs):
        cmd = [cc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", LIB]
        if verbose:
            print("[build]", " ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    return LIB


def asan_runtime() -> str:
    """The AddressSanitizer runtime of hipcc's clang (to LD_PRELOAD into the python that loads the instrumented library)."""
    clang = os.path.join(os.path.dirname(os.path.realpath(hipcc())), "..", "lib", "llvm", "bin", "clang")
    if not os.path.exists(clang):
        clang = "/opt/rocm/lib/llvm/bin/clang"
    return subprocess.run([clang, "-print-file-name=libclang_rt.asan-x86_64.so"], capture_output=True, text=True).stdout.strip()


def build_asan(force: bool = False, verbose: bool = True) -> str:
    """The same sources with the HOST side instrumented (-Xarch_host -fsanitize=address,undefined): the sf_* argument checks, the
    launch planning (grid shaping, dispatch tables, workspace sizes) and the error paths run under the sanitizers on a CPU-only box
    (kernel launches there fail with 'no device' AFTER the planning code has run).  Device code is unchanged: GPU AddressSanitizer
    is not available on this pool.  -> streamflow_amd/libstreamflow_hip_asan.so; tests/test_host_sanitizer_cpu.py"""
    os.makedirs(ASAN_OBJ, exist_ok=True)
    cc = hipcc()
    flags = ["--offload-arch=gfx950", "-O1", "-g", "-std=c++17", "-fPIC", "-fno-slp-vectorize", "-Wno-unused-function",
             "-Xarch_host", "-fsanitize=address,undefined", "-Xarch_host", "-fno-omit-frame-pointer", "-Xarch_host", "-fno-sanitize-recover=undefined"]
    jobs = []
    for src in SOURCES:
        s, o = os.path.join(CSRC, src), os.path.join(ASAN_OBJ, src.replace(".hip", ".o"))
        if force or _stale(o, [s] + HEADERS):
            jobs.append((s, o))

    def one(job):
        s, o = job
        cmd = [cc] + flags + ["-c", s, "-o", o]
        if verbose:
            print("[build-asan]", " ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc (asan) failed for {s}:\n{r.stderr[-4000:]}")
        return o

    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(one, jobs))
    objs = [os.path.join(ASAN_OBJ, s.replace(".hip", ".o")) for s in SOURCES]
    if force or jobs or _stale(ASAN_LIB, objs):
        r = subprocess.run([cc, "--offload-arch=gfx950", "-shared", "-fPIC", "-fsanitize=address,undefined", "-shared-libsan"] + objs +
                           ["-o", ASAN_LIB], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link (asan) failed:\n{r.stderr[-4000:]}")
    return ASAN_LIB


if __name__ == "__main__":
    if "--asan" in sys.argv:
        print(build_asan(force="--force" in sys.argv))
        sys.exit(0)
    if "--clean" in sys.argv:
        clean()
    print(build(force="--force" in sys.argv))
    if "--resources" in sys.argv:
        print(resources_markdown())
    check_resources()
