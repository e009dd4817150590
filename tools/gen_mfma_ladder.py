"""Generates tools/mfma_ladder.hip: a ladder of hand-scheduled inner loops (fixed physical registers,
one asm block each) from a bare fp64 MFMA loop up to the GEMM's full slab pipeline, to find which
ingredient costs MFMA issue slots on gfx950.  Run the binary on the GPU box.

register map (per wave): v[0:127] 16 accumulators (8 VGPRs each); v[128:143] fragment set 0
(a0..a3 b0..b3, 2 VGPRs each); v[144:159] fragment set 1; v[160:191] staging for ds_write /
global_load (8 x 4 VGPRs); v200 LDS read address, v201 LDS write address, v[202:203] global address.
"""
import sys

def mfma(acc, a, b):
    return f"v_mfma_f64_16x16x4_f64 v[{acc*8}:{acc*8+7}], v[{a}:{a+1}], v[{b}:{b+1}], v[{acc*8}:{acc*8+7}]"

def kstep_mfmas(fs):
    base = 128 + 16 * fs
    out = []
    for i in range(4):
        for j in range(4):
            out.append(mfma(i * 4 + j, base + 2 * i, base + 8 + 2 * j))
    return out

def reads(fs, kind, koff):
    """LDS reads of one k-step's 8 fragments into fragment set fs. koff: byte offset of the k-step."""
    base = 128 + 16 * fs
    out = []
    if kind == "read2":   # 4 x ds_read2_b64 (two row blocks 16 rows apart, stride 16 elements of 8 bytes)
        for q in range(4):
            off0 = (koff // 8) + 32 * (q % 2) + (64 if q >= 2 else 0)
            out.append(f"ds_read2_b64 v[{base+4*q}:{base+4*q+3}], v200 offset0:{off0 % 256} offset1:{(off0+16) % 256}")
    elif kind == "read1":  # 8 x ds_read_b64
        for q in range(8):
            out.append(f"ds_read_b64 v[{base+2*q}:{base+2*q+1}], v200 offset:{koff + 2176*q % 32768}")
    return out

def body(variant):
    L = []
    rd = variant.get("reads")
    for ks in range(4):
        fs = ks & 1
        ms = kstep_mfmas(fs)
        if variant.get("barrier") == "ks2" and ks == 3:
            L.append("s_barrier")
        if variant.get("spread") and ks in (2, 3):
            extra = []
            if ks == 2:
                L.append("s_waitcnt vmcnt(0)")
                extra = [f"ds_write_b128 v201, v[{160+4*q}:{163+4*q}] offset:{q*2048}" for q in range(8)]
            else:
                L += ["s_add_u32 s24, s24, 0x80000", "s_addc_u32 s25, s25, 0", "s_add_u32 s26, s26, 0x80000", "s_addc_u32 s27, s27, 0",
                      "s_add_u32 s21, s21, 1", "s_cmp_lt_u32 s21, 240", "s_cselect_b32 s24, s24, s28", "s_cselect_b32 s25, s25, s29",
                      "s_cselect_b32 s26, s26, s30", "s_cselect_b32 s27, s27, s31", "s_cselect_b32 s21, s21, 0"]
                extra = [f"global_load_dwordx4 v[{160+4*q}:{163+4*q}], v{204+q}, s[24:25]" for q in range(4)] + \
                        [f"global_load_dwordx4 v[{176+4*q}:{179+4*q}], v{204+q}, s[26:27]" for q in range(4)]
            r = reads(fs ^ 1, rd, 32 * ((ks + 1) % 4))
            L += r
            for i, m in enumerate(ms):
                L.append(m)
                if i % 2 == 1:
                    L.append(extra[i // 2])
            L.append("s_waitcnt lgkmcnt(0)")
        elif rd:
            r = reads(fs ^ 1, rd, 32 * ((ks + 1) % 4))
            if variant.get("interleave"):   # spread the reads between the MFMAs
                step = len(ms) // len(r)
                mixed = []
                for i, m in enumerate(ms):
                    if i % step == 0 and i // step < len(r):
                        mixed.append(r[i // step])
                    mixed.append(m)
                L += mixed
            else:
                L += r + ms
            L.append("s_waitcnt lgkmcnt(0)")
        else:
            L += ms
        if variant.get("mid") and ks == 1:
            L.append("s_waitcnt vmcnt(0)")
            for q in range(8):
                L.append(f"ds_write_b128 v201, v[{160+4*q}:{163+4*q}] offset:{q*2048}")
        if ks == (1 if variant.get("mid") else 0) and variant.get("gload") == "stream":
            # k-major panels like lauum: 16 rows of a 4096-wide matrix per slab, two operands (s[24:25], s[26:27])
            L.append("s_add_u32 s24, s24, 0x80000")   # 16 rows * 4096 * 8 bytes
            L.append("s_addc_u32 s25, s25, 0")
            L.append("s_add_u32 s26, s26, 0x80000")
            L.append("s_addc_u32 s27, s27, 0")
            L.append("s_add_u32 s21, s21, 1")
            L.append("s_cmp_lt_u32 s21, 240")          # wrap before the end of the 4096-row sample
            L.append("s_cselect_b32 s24, s24, s28")
            L.append("s_cselect_b32 s25, s25, s29")
            L.append("s_cselect_b32 s26, s26, s30")
            L.append("s_cselect_b32 s27, s27, s31")
            L.append("s_cselect_b32 s21, s21, 0")
            for q in range(4):
                L.append(f"global_load_dwordx4 v[{160+4*q}:{163+4*q}], v{204+q}, s[24:25]")
            for q in range(4):
                L.append(f"global_load_dwordx4 v[{176+4*q}:{179+4*q}], v{204+q}, s[26:27]")
        elif ks == 0 and variant.get("gload") is True:
            for q in range(8):
                L.append(f"global_load_dwordx4 v[{160+4*q}:{163+4*q}], v[202:203], off offset:{q*512}")
    if variant.get("gload") and not variant.get("mid"):
        L.append("s_waitcnt vmcnt(0)")
    if variant.get("writes") and not variant.get("mid"):
        for q in range(8):
            L.append(f"ds_write_b128 v201, v[{160+4*q}:{163+4*q}] offset:{q*2048}")
        L.append("s_waitcnt lgkmcnt(0)")
    if variant.get("barrier") is True:
        L.append("s_barrier")
    return L

VARIANTS = [
    ("mfma only", {}),
    ("+ds_read2 x4 per k-step (ahead of the MFMAs)", {"reads": "read2"}),
    ("+ds_read2 interleaved with MFMAs", {"reads": "read2", "interleave": True}),
    ("+ds_read_b64 x8 per k-step", {"reads": "read1"}),
    ("read2 + barrier per slab", {"reads": "read2", "barrier": True}),
    ("read2 + ds_write_b128 x8 + barrier", {"reads": "read2", "writes": True, "barrier": True}),
    ("read2 + global_load x8 + ds_write x8 + barrier (full slab pipeline)", {"reads": "read2", "gload": True, "writes": True, "barrier": True}),
    ("full slab pipeline, global loads streaming 16-row panels of 4096-wide matrices", {"reads": "read2", "gload": "stream", "writes": True, "barrier": True}),
    ("streaming, LDS writes + next loads issued mid-slab (after k-step 1)", {"reads": "read2", "gload": "stream", "writes": True, "barrier": True, "mid": True}),
    ("same with interleaved fragment reads", {"reads": "read2", "interleave": True, "gload": "stream", "writes": True, "barrier": True, "mid": True}),
    ("mid-slab writes, barrier after k-step 2 (next slab's first fragments prefetched during k-step 3)", {"reads": "read2", "gload": "stream", "writes": True, "barrier": "ks2", "mid": True}),
    ("LDS writes spread over k-step 2's MFMAs, global loads over k-step 3's", {"reads": "read2", "spread": True, "barrier": True}),
    ("streaming global loads only", {"gload": "stream"}),
    ("barrier only", {"barrier": True}),
    ("global_load x8 only", {"gload": True}),
]

def kernel(idx, variant):
    lines = body(variant)
    asm = []
    asm.append("s_mov_b32 s20, %1")
    asm.append("v_mov_b32 v200, %2")
    asm.append("v_mov_b32 v201, %3")
    asm.append("v_mov_b32 v202, %4")
    asm.append("v_mov_b32 v203, %5")
    for q in range(4):
        asm.append(f"v_mov_b32 v{204+q}, %{6+q}")
    asm.append("s_mov_b32 s24, %10")
    asm.append("s_mov_b32 s25, %11")
    asm.append("s_mov_b32 s26, %12")
    asm.append("s_mov_b32 s27, %13")
    asm.append("s_mov_b32 s28, s24")
    asm.append("s_mov_b32 s29, s25")
    asm.append("s_mov_b32 s30, s26")
    asm.append("s_mov_b32 s31, s27")
    asm.append("s_mov_b32 s21, 0")
    for r in range(0, 192):
        asm.append(f"v_mov_b32 v{r}, 0")
    asm.append("LOOP%=:")
    asm += lines
    asm.append("s_sub_u32 s20, s20, 1")
    asm.append("s_cmp_lg_u32 s20, 0")
    asm.append("s_cbranch_scc1 LOOP%=")
    asm.append("s_nop 7")
    asm.append("v_mov_b32 %0, v0")
    text = "\n".join(f'      "{a}\\n"' for a in asm)
    clob = ", ".join(f'"v{r}"' for r in range(0, 208)) + ", " + ", ".join(f'"s{r}"' for r in range(21, 32))
    return f"""
__global__ __launch_bounds__(256) void k{idx}(float* out, int iters, const double* gsrc, const double* big) {{
  extern __shared__ double lds[];
  for (int i = threadIdx.x; i < 8192; i += 256) lds[i] = 1e-3 * (i % 97);
  __syncthreads();
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  int raddr = ((lane & 15) * 17 + (lane >> 4)) * 8 + w * 64;
  int waddr = threadIdx.x * 16 + 32768;
  const double* gp = gsrc + (size_t)(blockIdx.x % 64) * 65536 + threadIdx.x * 2;
  unsigned lo = (unsigned)(size_t)gp, hi = (unsigned)((size_t)gp >> 32);
  // streaming operands: sample = block % 16 (4096 x 4096 doubles each), column panels of 128
  const double* pa = big + (size_t)(blockIdx.x % 16) * 4096 * 4096 + ((blockIdx.x / 16) % 32) * 128;
  const double* pb = big + (size_t)(blockIdx.x % 16) * 4096 * 4096 + ((blockIdx.x / 16 + 7) % 32) * 128;
  unsigned vo0 = ((threadIdx.x / 64 + 0) * 4096 + (threadIdx.x % 64) * 2) * 8, vo1 = vo0 + 4 * 4096 * 8, vo2 = vo0 + 8 * 4096 * 8,
           vo3 = vo0 + 12 * 4096 * 8;
  unsigned alo = __builtin_amdgcn_readfirstlane((unsigned)(size_t)pa), ahi = __builtin_amdgcn_readfirstlane((unsigned)((size_t)pa >> 32));
  unsigned blo = __builtin_amdgcn_readfirstlane((unsigned)(size_t)pb), bhi = __builtin_amdgcn_readfirstlane((unsigned)((size_t)pb >> 32));
  float r;
  asm volatile(
{text}
      : "=v"(r) : "s"(iters), "v"(raddr), "v"(waddr), "v"(lo), "v"(hi), "v"(vo0), "v"(vo1), "v"(vo2), "v"(vo3), "s"(alo), "s"(ahi), "s"(blo), "s"(bhi) : {clob}, "s20", "scc", "memory");
  out[blockIdx.x * 256 + threadIdx.x] = r;
}}
"""

src = ["// generated by tools/gen_mfma_ladder.py -- do not edit", "#include <hip/hip_runtime.h>", "#include <cstdio>"]
for i, (name, v) in enumerate(VARIANTS):
    src.append(kernel(i, v))
src.append("""
typedef void (*kern_t)(float*, int, const double*, const double*);
int main() {
  float* d; hipMalloc(&d, 4096 * 256 * 4);
  double* g; hipMalloc(&g, 64 * 65536 * 8 + (1 << 20)); hipMemset(g, 0, 64 * 65536 * 8 + (1 << 20));
  double* big; size_t bigsz = (size_t)16 * 4096 * 4096 * 8 + (8 << 20);
  if (hipMalloc(&big, bigsz) != hipSuccess) { printf("big alloc failed\\n"); return 1; }
  hipMemset(big, 0, bigsz);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 4000;  // slabs of 64 MFMAs
  struct { kern_t k; const char* name; } ks[] = {""")
for i, (name, v) in enumerate(VARIANTS):
    src.append(f'    {{k{i}, "{name}"}},')
src.append("""  };
  for (auto& e : ks) {
    for (int wgs : {256, 512}) {
      const unsigned dyn = 72 * 1024;  // two blocks per CU, like the GEMM
      hipFuncSetAttribute((const void*)e.k, hipFuncAttributeMaxDynamicSharedMemorySize, dyn);
      hipLaunchKernelGGL(e.k, dim3(wgs), dim3(256), dyn, 0, d, 50, g, big);
      hipDeviceSynchronize();
      hipEventRecord(e0);
      hipLaunchKernelGGL(e.k, dim3(wgs), dim3(256), dyn, 0, d, iters, g, big);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      hipError_t err = hipGetLastError();
      double tf = (double)wgs * 4 * 64.0 * iters * 2048 / (ms * 1e-3) / 1e12;
      printf("%-70s %d blocks/CU: %7.3f ms  %6.2f TFLOP/s%s\\n", e.name, wgs / 256, ms, tf, err ? "  (ERROR)" : "");
    }
  }
  return 0;
}
""")
open(sys.argv[1] if len(sys.argv) > 1 else "tools/mfma_ladder.hip", "w").write("\n".join(src))
