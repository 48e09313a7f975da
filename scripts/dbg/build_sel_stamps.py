#!/usr/bin/env python
"""Builds scripts/ab_libs_prof/libeqf_hip.so with time stamps (s_memrealtime) inside k_stats_select: kernel entry, evaluation done (both halves), barrier, ranking done, barrier,
stores done; EQF_DBG_SEL=1 prints the averages when the process exits. The tree is left as it was.
usage: python scripts/dbg/build_sel_stamps.py && EQF_DBG_SEL=1 EQVIO_AMD_LIB_DIR=scripts/ab_libs_prof python scripts/dbg/shipped_frames.py shipped 400"""
import os, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
src = os.path.join(ROOT, "eqvio_amd", "csrc")
tmp = "/tmp/sel_stamps_src"
shutil.rmtree(tmp, ignore_errors=True); shutil.copytree(src, tmp)
def edit(name, pairs):
    p = os.path.join(tmp, name); s = open(p).read()
    for old, new in pairs:
        assert s.count(old) == 1, (name, s.count(old), old)
        s = s.replace(old, new)
    open(p, "w").write(s)
T = "__builtin_amdgcn_s_memrealtime()"
edit("eqf_kernels.hpp", [
    ("double thrProb, int max_outliers, int M, int* __restrict__ removed_host, int* __restrict__ live_cols) {\n    __shared__ signed char s_kind[SEL_ONE_WG];",
     "double thrProb, int max_outliers, int M, int* __restrict__ removed_host, int* __restrict__ live_cols, unsigned long long* __restrict__ dbg = nullptr) {\n    if (dbg && threadIdx.x == 0) dbg[0] = " + T + ";\n    __shared__ signed char s_kind[SEL_ONE_WG];"),
    ("            s_cand[c] = SelCand{isabs ? ae : pe, isabs ? 2 : 1, i};\n        }\n    }\n    __syncthreads();\n",
     "            s_cand[c] = SelCand{isabs ? ae : pe, isabs ? 2 : 1, i};\n        }\n    }\n    if (dbg && (tid == 0 || tid == 256)) dbg[tid == 0 ? 1 : 2] = " + T + ";\n    __syncthreads();\n    if (dbg && tid == 0) dbg[3] = " + T + ";\n"),
    ("        if (lane == 0 && nrm)\n            atomicAdd(&s_cnt[1], nrm);\n    }\n    __syncthreads();\n",
     "        if (lane == 0 && nrm)\n            atomicAdd(&s_cnt[1], nrm);\n    }\n    if (dbg && tid == 0) dbg[4] = " + T + ";\n    __syncthreads();\n    if (dbg && tid == 0) { dbg[5] = " + T + "; dbg[8] = (unsigned long long)s_cnt[0]; }\n"),
    ("            lmidx_dev[j] = i;\n        }\n    }\n}\n\n// ---------------------------------------------------------------------------------------------------\n// K8b",
     "            lmidx_dev[j] = i;\n        }\n    }\n    if (dbg) { asm volatile(\"s_waitcnt vmcnt(0)\" ::: \"memory\"); if (tid == 0 || tid == 256) dbg[tid == 0 ? 6 : 7] = " + T + "; }\n}\n\n// ---------------------------------------------------------------------------------------------------\n// K8b"),
])
edit("eqf_hip.hip", [
    ("HIPCHK(hipHostMalloc(&c->h_sel, sizeof(int) * ((size_t)c->Ncap + 2)));", "HIPCHK(hipHostMalloc(&c->h_sel, sizeof(int) * ((size_t)c->Ncap + 2 + 64)));"),
    ("thrAbs, thrProb, max_outliers, M, c->h_sel, live_first ? c->d_spec + 2 : (int*)nullptr);\n            HIPCHK(hipGetLastError());\n        } else {",
     "thrAbs, thrProb, max_outliers, M, c->h_sel, live_first ? c->d_spec + 2 : (int*)nullptr, getenv(\"EQF_DBG_SEL\") ? (unsigned long long*)(c->h_sel + ((c->Ncap + 2 + 15) & ~15)) : nullptr);\n            HIPCHK(hipGetLastError());\n        } else {"),
    ("        copy_stats();\n        ++c->sel_frames;", """        copy_stats();
        if (getenv("EQF_DBG_SEL")) {
            static double acc[9]; static long cnt = 0;
            struct P { ~P() { for (int k = 1; k < 8; ++k) fprintf(stderr, "[sel] stamp %d: +%.2f us\\n", k, acc[k] / cnt); fprintf(stderr, "[sel] candidates %.1f\\n", acc[8] / cnt); } };
            static P pr;
            const unsigned long long* d = (const unsigned long long*)(c->h_sel + ((c->Ncap + 2 + 15) & ~15));
            for (int k = 1; k < 8; ++k) acc[k] += 0.01 * (double)(long long)(d[k] - d[0]);
            acc[8] += (double)d[8];
            ++cnt;
        }
        ++c->sel_frames;"""),
])
out = os.path.join(ROOT, "scripts", "ab_libs_prof"); os.makedirs(out, exist_ok=True)
inc = os.path.join(ROOT, "include")
subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-I" + inc, "-Wno-unused", "-Wno-pass-failed", "-shared", "-o", os.path.join(out, "libeqf_hip.so"), os.path.join(tmp, "eqf_hip.hip")])
host = os.path.join(ROOT, "eqvio_amd", "host")
subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-Wno-unused", "-I" + inc, "-I/opt/rocm/include", "-D__HIP_PLATFORM_AMD__", "-shared", "-o", os.path.join(out, "libeqvio_filter.so")] +
                      [os.path.join(host, f) for f in ("VIOFilter.cpp", "VIOSimulator.cpp", "VIOWriter.cpp", "DatasetReplay.cpp", "filter_capi.cpp", "sim_capi.cpp")] + ["-L" + out, "-leqf_hip", "-Wl,-rpath,$ORIGIN"])
print("built", out)
