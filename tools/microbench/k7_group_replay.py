"""Development tool: the issue ceiling of K7's OWN instruction mix.

Takes the 4-entry group body of `blend_backward_kernel<0, false, false>` exactly as hipcc emits it for the product
(`gaussianeditor_amd/csrc/gsr_blend.hip`, the inner loop of backward_tile: footprints + exp of four entries, the
transmittance / colour recurrences, the nine moments, the 4-entry transposed wave reduction and the nine ds_add_f32),
strips its control flow (the `no lane contributes` skip and the loop's own back edge), and replays it REP times per wave
from zero-initialised registers and LDS -- no global memory, no barrier, no list walk.  Launched as 4-wave workgroups (one
wave per SIMD, like the product) with 1 .. 4 workgroups per CU on every CU, it measures how many cycles a SIMD needs per
group when 1 .. 4 waves share it: THAT, not a per-instruction figure, is the ceiling K7's `valu_issue_frac` is quoted against
(bench.py: K7_GROUP_CYCLES_PER_SIMD).  Variants: `full` (as compiled), `nolds` (every ds_* instruction and the waits for
them removed: the VALU/SALU stream alone), `noxlane` (additionally without the permlane / DPP reduction steps).

    python tools/microbench/k7_group_replay.py            # writes build_variants/k7_group_replay.hip and builds it (no GPU needed)
    build_variants/k7_group_replay                        # on the MI355X: prints one table (and JSON with --json)
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
OUT = os.path.join(ROOT, "build_variants")
KERNEL = "_ZN3gsr21blend_backward_kernelILi0ELb0ELb0EEEvNS_9BlendArgsE"
FLAGS = ("--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics -fno-gpu-rdc -fno-slp-vectorize "
         "-fvisibility=hidden -mllvm -amdgpu-sched-strategy=max-ilp -mllvm -amdgpu-atomic-optimizer-strategy=None").split()


def product_isa(extra=()) -> str:
    os.makedirs(OUT, exist_ok=True)
    s = os.path.join(OUT, "gsr_blend.s")
    subprocess.check_call(["/opt/rocm/bin/hipcc", *FLAGS, *extra, "-S", "--cuda-device-only", "-o", s,
                           os.path.join(ROOT, "gaussianeditor_amd", "csrc", "gsr_blend.hip")], stderr=subprocess.DEVNULL)
    return open(s).read()


def group_body(isa: str):
    """The first depth-2 loop of the kernel that holds the transposed reduction: lines from its header label to the jump
    back, in program order, plus the two join blocks behind it (exec restore, loop increment)."""
    lines = isa.split("\n")
    a = next(i for i, ln in enumerate(lines) if ln.startswith(KERNEL + ":"))
    b = next(i for i in range(a, len(lines)) if lines[i].startswith("\t.end_amdhsa_kernel") or lines[i].startswith(".Lfunc_end"))
    k = lines[a:b]
    heads = [i for i, ln in enumerate(k) if "Inner Loop Header: Depth=2" in ln]
    is_label = lambda ln: re.match(r"^\.LBB\d+_\d+:", ln) is not None  # noqa: E731
    for h in heads:
        j = h
        while not is_label(k[j]):  # the loop's label: the header comment's own line, or the one in front of it
            j -= 1
        end = next(i for i in range(j + 1, len(k)) if is_label(k[i]) and "Depth=2" not in k[i])
        # the loop's join blocks (exec restore, increment + back edge) are emitted IN FRONT of the header: the run of
        # labels marked "Depth=2" that ends at the header
        start = j
        for i in range(j - 1, -1, -1):
            if is_label(k[i]):
                if "Depth=2" not in k[i]:
                    break
                start = i
        if any("v_permlane32_swap" in ln for ln in k[j:end]):
            return k[j].split(":")[0], k[start:j], k[j:end]
    raise SystemExit("group loop not found in the ISA")


RARE_MARK = "polynomial exp"  # the asm comment inside the backward's rarely taken exact-exponential block (gsr_blend.hip)


def drop_rare_blocks(lines):
    """Removes the fall-through block that holds RARE_MARK (from the `; %bb.N:` comment that opens it to the `s_branch`
    that closes it): the replay takes the common path, on which the conditional branch in front of it is taken."""
    marks = [i for i, ln in enumerate(lines) if RARE_MARK in ln]
    for m in reversed(marks):
        a = max(i for i in range(m) if lines[i].lstrip().startswith("; %bb."))
        b = next(i for i in range(m, len(lines)) if lines[i].split(";")[0].strip().startswith("s_branch"))
        lines = lines[:a] + lines[b + 1:]
    return lines


def clean(lines, variant):
    out = []
    for ln in drop_rare_blocks(list(lines)):
        t = ln.split(";")[0].rstrip()
        if not t.strip() or t.strip().startswith("."):
            continue
        t = t.strip()
        op = t.split()[0]
        if op in ("s_cbranch_vccz", "s_cbranch_vccnz", "s_cbranch_execz", "s_cbranch_execnz", "s_cbranch_scc1", "s_cbranch_scc0", "s_branch"):
            continue  # straight line: every group is evaluated in full, the masked tail included
        if op == "v_add_u32_e32" and t.replace(" ", "").startswith("v_add_u32_e32v2,64,v2"):
            t = "v_add_u32_e32 v2, 0, v2"  # (the LDS cursor of the loop: stays on the first group's records)
        if variant in ("nolds", "noxlane"):
            if op.startswith("ds_"):
                continue
            if op == "s_waitcnt" and "lgkmcnt" in t and "vmcnt" not in t:
                continue
        if variant == "noxlane" and (op.startswith("v_permlane") or op.endswith("_dpp")):
            continue
        if op == "s_waitcnt" and "vmcnt" in t and "lgkmcnt" not in t:
            continue  # (no vector memory in the replay)
        out.append(t)
    return out


def registers(body):
    v, s = set(), set()
    for t in body:
        for m in re.finditer(r"\bv\[(\d+):(\d+)\]", t):
            v.update(range(int(m.group(1)), int(m.group(2)) + 1))
        for m in re.finditer(r"\bv(\d+)\b", t):
            v.add(int(m.group(1)))
        for m in re.finditer(r"\bs\[(\d+):(\d+)\]", t):
            s.update(range(int(m.group(1)), int(m.group(2)) + 1))
        for m in re.finditer(r"\bs(\d+)\b", t):
            s.add(int(m.group(1)))
    return sorted(v), sorted(s)


def emit(isa: str) -> str:
    label, join, loop = group_body(isa)
    variants = {}
    for name in ("full", "nolds", "noxlane"):
        body = clean(loop, name) + clean(join, name)
        variants[name] = body
    v, s = registers(variants["full"])
    cnt = max(r for r in range(96) if r not in s)  # loop counter SGPR, outside the body's set (s100+ are reserved)
    src = ['// GENERATED by tools/microbench/k7_group_replay.py from the product\'s ISA -- do not edit.',
           '#include <hip/hip_runtime.h>', '#include <cstdint>', '#include <cstdio>', '#include <cstring>', '#include <vector>', '']
    for name, body in variants.items():
        n_valu = sum(1 for t in body if t.startswith("v_"))
        n_salu = sum(1 for t in body if t.startswith("s_") and not t.startswith("s_waitcnt"))
        n_lds = sum(1 for t in body if t.startswith("ds_"))
        init = [f"v_mov_b32 v{r}, 0" for r in v] + [f"s_mov_b32 s{r}, 0" for r in s if r not in (8, 9)]
        # s[8:9]: the lanes that own a row total (lane & 15 == 15), what the masked tail runs under
        init += ["s_mov_b32 s8, 0x80008000", "s_mov_b32 s9, 0x80008000", f"s_mov_b32 s{cnt}, %0"]
        text = init + [f"K7R_{name}_%=:"] + body + [f"s_sub_u32 s{cnt}, s{cnt}, 1", f"s_cmp_lg_u32 s{cnt}, 0", f"s_cbranch_scc1 K7R_{name}_%="]
        asm = "\n".join(f'      "{t}\\n"' for t in text)
        clob = ", ".join([f'"v{r}"' for r in v] + [f'"s{r}"' for r in s] + [f'"s{cnt}"', '"vcc"', '"scc"', '"memory"'])
        src += [f"// {name}: {n_valu} VALU + {n_salu} SALU + {n_lds} LDS instructions per group of four entries",
                f"constexpr int N_{name}[3] = {{{n_valu}, {n_salu}, {n_lds}}};",
                f"__global__ void __launch_bounds__(256) replay_{name}(uint64_t* cyc, int rep) {{",
                "  __shared__ float lds[5120];  // 20 KB, as the product's workgroup; zero: slot 0, null records",
                "  for (int i = threadIdx.x; i < 5120; i += 256) lds[i] = 0.f;",
                "  __syncthreads();",
                "  const uint64_t t0 = __builtin_amdgcn_s_memtime();",
                "  asm volatile(", asm, f"      : : \"s\"(rep) : {clob});",
                "  const uint64_t t1 = __builtin_amdgcn_s_memtime();",
                "  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;",
                "  if (rep < 0) lds[threadIdx.x] = 1.f;", "}", ""]
    src += [r'''
template <class K>
static void run(const char* name, K kern, const int n[3], int cus, bool json) {
  const int rep = 4000;
  for (int wgs = 1; wgs <= 5; ++wgs) {
    const int grid = cus * wgs;
    uint64_t* cyc;
    (void)hipMalloc(&cyc, sizeof(uint64_t) * grid * 4);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, cyc, rep);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, cyc, rep);
    (void)hipDeviceSynchronize();
    std::vector<uint64_t> h(grid * 4);
    (void)hipMemcpy(h.data(), cyc, sizeof(uint64_t) * grid * 4, hipMemcpyDeviceToHost);
    double mean = 0;
    uint64_t mx = 0;
    for (auto v : h) { mean += (double)v; mx = v > mx ? v : mx; }
    mean /= (double)h.size();
    const double per_wave = mean / rep, per_simd = per_wave / wgs, instr = n[0] + n[1] + n[2];
    if (json)
      printf("{\"variant\": \"%s\", \"waves_per_simd\": %d, \"cycles_per_group_wave\": %.1f, \"cycles_per_group_simd\": %.1f, "
             "\"cycles_per_instr_simd\": %.3f, \"valu\": %d, \"salu\": %d, \"lds\": %d, \"max_over_mean\": %.3f}\n",
             name, wgs, per_wave, per_simd, per_simd / instr, n[0], n[1], n[2], (double)mx / rep / per_wave);
    else
      printf("%-8s %d waves/SIMD: %8.1f cycles per group and wave, %7.1f per group and SIMD = %5.2f cycles per instruction "
             "(%d VALU + %d SALU + %d LDS), slowest wave %.3f x mean\n",
             name, wgs, per_wave, per_simd, per_simd / instr, n[0], n[1], n[2], (double)mx / rep / per_wave);
    fflush(stdout);
    (void)hipFree(cyc);
  }
}

int main(int argc, char** argv) {
  const bool json = argc > 1 && !strcmp(argv[1], "--json");
  hipDeviceProp_t p;
  (void)hipGetDeviceProperties(&p, 0);
  const int cus = p.multiProcessorCount;
  if (!json) printf("%s, %d CUs; 4-wave workgroups (one wave per SIMD), 1 .. 5 workgroups per CU, 4000 groups per wave\n", p.name, cus);
  run("full", replay_full, N_full, cus, json);
  run("nolds", replay_nolds, N_nolds, cus, json);
  run("noxlane", replay_noxlane, N_noxlane, cus, json);
  return 0;
}
''']
    return "\n".join(src)


def main():
    # optional: NAME followed by extra compiler flags for the product source, e.g. `poly -DGSR_BWD_HYBRID_EXP=0` (A/B builds)
    name = "k7_group_replay" + ("_" + sys.argv[1] if len(sys.argv) > 1 else "")
    isa = product_isa(sys.argv[2:])
    hip = os.path.join(OUT, name + ".hip")
    open(hip, "w").write(emit(isa))
    exe = os.path.join(OUT, name)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-Wno-unused-value", hip, "-o", exe])
    print("built", exe)


if __name__ == "__main__":
    main()
