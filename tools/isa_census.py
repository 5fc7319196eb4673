#!/usr/bin/env python3
"""Cost-weighted ISA census of one instantiation of the render kernel (no GPU needed): profiles/r05_isa_cost_census.txt.

For every instruction of the compiled kernel: its encoding class, its place in the SOURCE through the inline stack the compiler
recorded (llvm-symbolizer --inlines on a -gline-tables-only build: codegen is unchanged), hence its PHASE of the bounce loop, and
an execution frequency per wave-bounce from the measured trip counts of that phase (profiles/r04_block_counts*.txt; the per-phase
rules are in PHASES below).  static count x frequency x measured cycles per encoding (profiles/r04_valu_cost_table.txt, four waves
per SIMD) = modelled SIMD cycles per phase, to be read against the stamped shares (profiles/r04_phase_shares.txt) and against the
hardware's instruction counters (SQ_INSTS_VALU per wave-bounce).

usage: tools/isa_census.py [--kernel SUBSTR] [--tenk] [-D...]   default kernel: render_kernelILi5ELb0ELb1ELb0ELi256 (shipped, small grid)
"""
import collections, json, os, re, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
SRC = os.path.join(ROOT, "rtiow_amd", "csrc")
TMP = "/tmp/_isa_census"


def build(extra):
    os.makedirs(TMP, exist_ok=True)
    obj, elf = f"{TMP}/dev.o", f"{TMP}/dev.elf"
    subprocess.run(["hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-fhip-fp32-correctly-rounded-divide-sqrt",
                    "-mllvm", "-amdgpu-mfma-vgpr-form", "-I", "include", "-I", "rtiow_amd/csrc", "--cuda-device-only", "-c",
                    "-gline-tables-only", *extra, "-o", obj, "rtiow_amd/csrc/rt_api.hip"], cwd=ROOT, check=True, stderr=subprocess.DEVNULL)
    subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={obj}", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                    f"--output={elf}"], check=True)
    dis = subprocess.run([f"{LLVM}/llvm-objdump", "-d", "--no-show-raw-insn", elf], check=True, capture_output=True, text=True).stdout
    return elf, dis


def kernel_instructions(dis, want):
    """[(address, opcode, operands)] of the first function whose mangled name contains `want`."""
    out, inside = [], False
    for line in dis.splitlines():
        m = re.match(r"^([0-9a-f]+) <(\S+)>:", line)
        if m:
            if inside:
                break
            inside = want in m.group(2)
            continue
        if not inside:
            continue
        m = re.match(r"^\s+([a-z][a-z0-9_]+)\s*(.*?)\s*//\s*([0-9A-F]+):", line)
        if m:
            out.append((int(m.group(3), 16), m.group(1), m.group(2)))
    return out


def symbolize(elf, addrs):
    """address -> [(function, file, line)] innermost first."""
    p = subprocess.run([f"{LLVM}/llvm-symbolizer", f"--obj={elf}", "--inlines", "--output-style=JSON"],
                       input="\n".join(hex(a) for a in addrs) + "\n", capture_output=True, text=True, check=True)
    res = {}
    for line in p.stdout.splitlines():
        if not line.strip():
            continue
        d = json.loads(line)
        res[int(d["Address"], 16)] = [(s["FunctionName"], os.path.basename(s["FileName"]), s["Line"]) for s in d["Symbol"]]
    return res


# ---- encoding classes and their measured cost (SIMD cycles per instruction at four waves per SIMD) ---------------------------
# profiles/r04_valu_cost_table.txt (tools/valu_cost_table.hip); classes not in the table take the nearest measured relative.
def classify(op, args):
    sgpr_src = False
    if op.startswith("v_"):
        # a scalar or literal SOURCE operand (not the vcc/sgpr destination of a compare, not an inline constant)
        parts = [a.strip() for a in args.split(",")]
        srcs = parts[1:] if not op.startswith("v_cmp") else parts[1:]
        if op.startswith(("v_mad_u64_u32", "v_mad_i64_i32")):
            srcs = parts[2:]                                    # (vdst, sdst carry-out, src0, src1, src2)
        sgpr_src = any(re.match(r"^-?\|?s(\d+|\[)", a) or a in ("vcc", "exec") or re.match(r"^0x[0-9a-f]+$", a) for a in srcs
                       if not a.startswith(("bitop3", "offset", "row_", "quad_", "bank_", "bound_", "src", "op_sel", "neg", "clamp", "mul:", "div:")))
    if op.startswith("v_mfma"):
        return "mfma", 0.0
    if op.startswith("v_"):
        if op.startswith(("v_mad_u64_u32", "v_mad_i64_i32")):
            return "valu mad64" + (" +s" if sgpr_src else ""), 5.09 if sgpr_src else 4.80
        if op.startswith(("v_fma_f64", "v_fmac_f64")):
            return "valu f64 fma", 6.23
        if op.startswith(("v_mul_f64", "v_add_f64", "v_ldexp_f64", "v_min_f64", "v_max_f64", "v_div_scale_f64", "v_div_fmas_f64",
                          "v_div_fixup_f64", "v_frexp", "v_trunc_f64", "v_floor_f64", "v_rndne_f64", "v_fract_f64")):
            return "valu f64" + (" +s" if sgpr_src else ""), 4.9 if sgpr_src else 4.7
        if op.startswith(("v_rcp_f64", "v_rsq_f64", "v_sqrt_f64")):
            return "valu f64 trans", 9.0
        if op.startswith(("v_cvt_f64", "v_cvt_u32_f64", "v_cvt_i32_f64", "v_cvt_f32_f64")):
            return "valu cvt64", 4.6
        if op.startswith(("v_cmp", "v_cmpx")):
            return "valu cmp" + ("64" if "64" in op.split("_")[-2:][0] or op.endswith(("_f64_e32", "_f64_e64", "_u64_e32", "_u64_e64", "_i64_e64", "_i64_e32")) else ""), 4.9
        if op.startswith(("v_rcp_f32", "v_rsq_f32", "v_sqrt_f32", "v_exp", "v_log", "v_sin", "v_cos")):
            return "valu trans32", 8.0
        if op.startswith(("v_max", "v_min", "v_med3")):
            return "valu minmax", 4.3
        if op.startswith(("v_readlane", "v_readfirstlane", "v_writelane")):
            return "valu lane", 4.4
        if op.startswith(("v_cndmask",)):
            return "valu cndmask", 4.44
        if op.startswith(("v_lshl_add", "v_add_lshl", "v_lshl_or", "v_and_or", "v_add3", "v_or3", "v_bfe", "v_bfi", "v_alignbit", "v_perm", "v_mad_u32", "v_mul_lo", "v_mul_hi", "v_mul_u32")):
            return "valu 3op/int" + (" +s" if sgpr_src else ""), 4.4 if sgpr_src else 4.23
        if op.startswith("v_bitop3"):
            return "valu bitop3" + (" +s" if sgpr_src else ""), 4.33 if sgpr_src else 3.18
        if "dpp" in args or op.endswith("_dpp"):
            return "valu dpp", 4.4
        return "valu plain" + (" +s" if sgpr_src else ""), 4.21 if sgpr_src else 2.9
    if op.startswith(("s_cbranch", "s_branch")):
        return "branch", 0.0
    if op.startswith(("s_waitcnt", "s_nop", "s_setprio", "s_sleep")):
        return "wait/nop", 0.0
    if op.startswith(("s_load", "s_buffer", "s_memtime", "s_memrealtime")):
        return "smem", 0.0
    if op.startswith("s_"):
        return "salu", 0.0
    if op.startswith("ds_"):
        return "lds", 0.0
    if op.startswith(("buffer_", "global_", "flat_", "scratch_")):
        return "vmem", 0.0
    return "other", 0.0


def source_lines():
    return open(os.path.join(SRC, "rt_kernels.hpp")).read().splitlines()


def find_line(lines, needle, start=0):
    for i in range(start, len(lines)):
        if needle in lines[i]:
            return i + 1
    raise SystemExit(f"anchor not found in rt_kernels.hpp: {needle!r}")


def phases_table(lines, counts):
    """[(name, first line, last line, executions per wave-bounce, static copies)]: by line of rt_kernels.hpp of ANY frame of the
    inline stack, the first match in this order wins (inner lambdas before the code that calls them).  Executions: measured trip
    counts (tools/block_counts.py) per wave-bounce; a phase that runs once per pass has 1.  The compiler REPLICATES code (the tile
    loop holds three copies of do_tile, each with a pipelined and a group-by-group body: 24 copies of the look, one per static
    v_mfma): the executions are spread over the copies, so one static instruction runs executions / copies times.  `copies` is a
    number or the name of a signature instruction whose static count in the phase gives it ("mfma": one per look copy; "philox":
    v_mad_u64_u32 / 20, one Philox block each)."""
    L = lambda s, st=0: find_line(lines, s, st)
    c = counts
    look0 = L("auto look_tube = [&]")
    keep0 = L("if (__builtin_expect((kh[0] | kh[1]) != 0ull, 0))", look0)
    half0 = L("if (kh[bb] != 0ull)", keep0)
    look_end = L("// ---- which tiles: the global ones and the grid cells some ray of the wave can reach", look0) - 1
    pool0 = L("auto pool_round = [&]()")
    pool_end = L("// owners push the candidates of segment seg0", pool0) - 1
    enum0 = L("auto enumerate = [&]")
    enum_loop = L("int wbase = 0;", enum0)
    enum_end = L("// after the last segment: drain the ring", enum0) - 1
    fin0 = enum_end + 1
    fin_end = L("const int nt = P.n_tiles;", fin0) - 1
    tube0 = L("if constexpr (TUBE) {", fin_end)
    always0 = L("// the spheres that skip the filter (the ground): tested exactly by every ray", tube0)
    seed0 = L("typedef float f32x16 __attribute__((ext_vector_type(16)));", always0)
    which0 = look_end + 1
    loop0 = L("for (int t0 = 0; t0 < n_list; t0 += kSeg / 2) {", which0)
    dotile0 = L("auto do_tile = [&]", loop0)
    pipe0 = L("// B operands ping-pong between two register sets, each fetched a tile ahead.", dotile0)
    loop_end = L("__builtin_amdgcn_s_setprio(1);", pipe0)
    take0 = L("// ---- (a) lanes without a path take the next camera rays of the wave's queue")
    refill0 = L("if (q_count == 0u) {", take0)
    refill_end = L("const uint32_t r = rank_below(m);", refill0) - 1
    cam0 = L("if (fresh) {", refill_end)
    hit0 = L("// ---- (c) every lane of the wave is out of work: done", cam0)
    exact0 = L("auto exact_test_g = [&]", hit0)
    exact_end = L("auto test_list = [&]", exact0) - 1
    shade0 = L("// ---- (e) shade: main.rs:44-56 + materials.rs")
    retry0 = L("if (kind != RT_KIND_DIALECTRIC) {", L("} else", L("RT_STAMP(11);", shade0)))
    retry_loop = L("do {", retry0)
    retry_end = L("ev += nblk;", retry_loop)
    unit0 = L("// Every branch normalises exactly one vector", retry_end)
    fin_s0 = L("// ---- (f) finished samples -> their block's sums", unit0)
    flush0 = L("auto flush_ring = [&]")
    end0 = L("// every block of this wave has finished its last sample", fin_s0)
    return [
        ("tile loop: keep path, per half (4 x or/cmp/ds_or)", half0, look_end, c["halves"], "2mfma"),     # two static halves per look copy
        ("tile loop: keep path, per look with a candidate", keep0, half0 - 1, c["keeps"], "mfma"),
        ("tile loop: MFMA + look (per 16-ray group x tile)", look0, keep0 - 1, c["looks"], "mfma"),
        ("pooled exact round (sphere.rs:16-34, 64 pairs)", pool0, pool_end, c["pool_rounds"], "ds_min"),
        ("enumerate: per trip of the bitmap walk", enum_loop, enum_end, c["enum_trips"], 1),
        ("enumerate: summary words", enum0, enum_loop - 1, 1.0, 1),
        ("finish pool / take the minimum", fin0, fin_end, 1.0, 1),
        ("ground sphere exact test (always-exact list)", exact0, exact_end, 1.0, 3),      # three inlined copies (first always-exact sphere, the loop over
                                                                                      # further ones, the in-order form for out-of-range rays): the first runs
        ("tile loop: per tile (B operand, list entry, dispatch)", dotile0, loop_end, c["tiles"], 3),
        ("tile loop: per segment (zero the words, list read)", loop0, dotile0 - 1, 1.0, 1),
        ("footprints + tile list", which0, loop0 - 1, 1.0, 1),
        ("filter rows: make_tube, bf16 pieces, staging", tube0, always0 - 1, 1.0, 1),
        ("always-exact list, out-of-range rays", always0, seed0 - 1, c["always_extra"], 1),   # loop control + the whole-list scan of a ray outside the
                                                                                          # filter's range: (all but) never on these scenes
        ("refill: start 64 samples (item -> pixel, Philox, lens)", refill0, refill_end, c["refills"], 1),
        ("take samples from the queue", take0, cam0 - 1, 1.0 + c["refills"], 1),
        ("camera ray of fresh lanes (camera.rs:47-54)", cam0, hit0 - 1, 1.0, 1),
        ("unit-sphere retry loop (3 static blocks + 4 tries per trip)", retry_loop, retry_end, c["retry_blocks"], "philox"),
        ("hit record, material fetch, first Philox block, try 0", shade0, retry_loop - 1, 1.0, 1),
        ("unit_vector + materials + next ray", unit0, fin_s0 - 1, 1.0, 1),
        ("block write-out (flush_ring)", flush0, L("};", flush0), c["flushes"], "flush"),
        ("finished samples: quantise, block sums, bookkeeping", fin_s0, end0 - 1, 1.0, 1),
        ("pass prologue (alive ballot, a = d.d)", hit0, exact0 - 1, 1.0, 1),
    ]


DEVICE_FUNCS = {  # functions of rt_device.hpp that get their own column in the per-phase split
    "philox4x32_10": "Philox block",
}


def main():
    args = sys.argv[1:]
    want = "render_kernelILi5ELb0ELb1ELb0ELi256"
    tenk = "--tenk" in args
    if "--kernel" in args:
        want = args[args.index("--kernel") + 1]
    elif tenk:
        want = "render_kernelILi5ELb0ELb0ELb0ELi1024"
    extra = [a for a in args if a.startswith("-D")]
    # measured trips per wave-bounce (profiles/r04_block_counts.txt: 1200x675x100; r04_block_counts_cfg4.txt: 10k spheres)
    counts = ({"looks": 20.0, "keeps": 14.03, "halves": 21.56, "pool_rounds": 1.90, "enum_trips": 5.6, "tiles": 8.43, "refills": 0.355,
               "retry_blocks": 3.8, "flushes": 0.36 / 16.0, "always_extra": 0.3}
              if tenk else
              {"looks": 15.22, "keeps": 10.18, "halves": 16.24, "pool_rounds": 1.62, "enum_trips": 4.6, "tiles": 5.35, "refills": 0.377,
               "retry_blocks": 3.72, "flushes": 0.377 / 4.0, "always_extra": 0.1})
    elf, dis = build(extra)
    ins = kernel_instructions(dis, want)
    if not ins:
        raise SystemExit(f"no kernel matching {want}")
    sym = symbolize(elf, [a for a, _, _ in ins])
    lines = source_lines()
    phases = phases_table(lines, counts)
    dump = args[args.index("--dump") + 1] if "--dump" in args else None     # print the instructions of the phases whose name contains this
    static = collections.defaultdict(collections.Counter)       # phase -> class -> static count
    cost = {}
    unplaced = collections.Counter()
    philox_in = collections.Counter()
    for addr, op, ops in ins:
        cls, cyc = classify(op, ops)
        cost[cls] = cyc
        stack = sym.get(addr, [])
        klines = [ln for fn, f, ln in stack if f == "rt_kernels.hpp"]
        ph = None
        for name, lo, hi, _, _ in phases:
            if any(lo <= ln <= hi for ln in klines):
                ph = name
                break
        if ph is None:
            ph = "outside the bounce loop / unplaced"
            unplaced[tuple(klines[:2])] += 1
        static[ph][cls] += 1
        if dump and dump in ph:
            print(f"{addr:08x} {cls:18s} {op} {ops}   ; {' < '.join(f'{f.split(chr(46))[0][-8:]}:{ln}' for fn, f, ln in stack[:3])}")
        if any(fn.startswith("rt::philox4x32_10") or "philox4x32_10" in fn for fn, f, ln in stack):
            philox_in[ph] += 1 if cls.startswith("valu") else 0
    n_mfma = sum(st.get("mfma", 0) for st in static.values())
    sig = collections.defaultdict(collections.Counter)          # phase -> signature opcode -> static count
    for addr, op, ops in ins:
        stack = sym.get(addr, [])
        klines = [ln for fn, f, ln in stack if f == "rt_kernels.hpp"]
        for name, lo, hi, _, _ in phases:
            if any(lo <= ln <= hi for ln in klines):
                sig[name][op] += 1
                break

    def copies_of(name, rule):
        if isinstance(rule, (int, float)):
            return float(rule)
        if rule == "mfma":
            return float(n_mfma)
        if rule == "2mfma":
            return 2.0 * n_mfma
        if rule == "philox":
            return max(1.0, sig[name]["v_mad_u64_u32"] / 20.0)
        if rule == "ds_min":
            return max(1.0, float(sig[name]["ds_min_rtn_u64"] + sig[name]["ds_min_u64"]))
        if rule == "flush":
            return max(1.0, float(sig[name]["global_atomic_add_x2"]))
        raise SystemExit(rule)

    classes = sorted({c for ph in static.values() for c in ph})
    vcls = [c for c in classes if c.startswith("valu")]
    print(f"kernel {want}: {len(ins)} instructions; executions per wave-bounce: {counts}")
    tot_static = collections.Counter()
    for ph in static.values():
        tot_static.update(ph)
    print("static totals:", {k: v for k, v in sorted(tot_static.items())})
    print()
    hdr = (f"{'phase':62s} {'exec':>6s} {'copies':>6s} {'sVALU':>6s} {'VALU/x':>6s} {'dVALU':>7s} {'dSALU':>6s} {'dLDS':>5s} {'cycles':>7s} {'share':>6s}"
           f"  of which Philox (static VALU)")
    print(hdr)
    rows, tot_v, tot_c, tot_s, tot_l = [], 0.0, 0.0, 0.0, 0.0
    freq = {}
    for name, lo, hi, ex, rule in phases + [("outside the bounce loop / unplaced", 0, 0, 0.0, 1)]:
        st = static.get(name, {})
        cp = copies_of(name, rule)
        f = ex / cp
        freq[name] = f
        sv = sum(st.get(c, 0) for c in vcls)
        dv = sv * f
        cyc = sum(st.get(c, 0) * cost[c] for c in vcls) * f
        ds_ = (st.get("salu", 0) + st.get("branch", 0)) * f
        dl = st.get("lds", 0) * f
        rows.append((name, ex, cp, sv, dv, ds_, dl, cyc, philox_in.get(name, 0)))
        tot_v += dv; tot_c += cyc; tot_s += ds_; tot_l += dl
    for name, ex, cp, sv, dv, ds_, dl, cyc, phx in rows:
        print(f"{name:62s} {ex:6.2f} {cp:6.0f} {sv:6d} {sv / cp:6.1f} {dv:7.1f} {ds_:6.1f} {dl:5.1f} {cyc:7.0f} {100 * cyc / max(tot_c, 1):5.1f}%  {phx}")
    print(f"{'MODEL TOTAL per wave-bounce':62s} {'':6s} {'':6s} {tot_v:7.1f} {tot_s:6.1f} {tot_l:5.1f} {tot_c:7.0f}")
    print()
    print("dynamic vector instructions by encoding class (modelled), cycles each (measured, 4 waves/SIMD), cycles per wave-bounce:")
    dyn = collections.Counter()
    for name, lo, hi, ex, rule in phases:
        for c in vcls:
            dyn[c] += static.get(name, {}).get(c, 0) * freq[name]
    for c, n in sorted(dyn.items(), key=lambda kv: -kv[1] * cost[kv[0]]):
        print(f"   {c:22s} {n:8.1f}  x {cost[c]:4.2f} = {n * cost[c]:7.0f}  ({100 * n * cost[c] / max(tot_c, 1):4.1f} %)")
    if "--by-phase-class" in args:
        print("\ndynamic count by phase x class (>= 3 per wave-bounce):")
        for name, lo, hi, ex, rule in phases:
            st = static.get(name, {})
            items = [(c, st.get(c, 0) * freq[name]) for c in vcls if st.get(c, 0) * freq[name] >= 3.0]
            if items:
                print(f"   {name[:50]:50s} " + "  ".join(f"{c[5:]}={n:.0f}" for c, n in sorted(items, key=lambda kv: -kv[1])))
    if unplaced:
        print("\nunplaced (kernel lines of the stack):", unplaced.most_common(12))


if __name__ == "__main__":
    main()
