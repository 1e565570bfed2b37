"""CPU tier (cross-compile only): ISA invariants the hand-scheduled kernels rely on.

gram_glds_kernel / gram_glds_f32_kernel (stats_kernels.hip) issue their LDS-DMA loads through inline asm - invisible to hipcc's
wait-count bookkeeping - and order the per-wave ring with hand-written `s_waitcnt vmcnt(N)`.  That count is right only while the
compiler puts NO other vector-memory instruction into the steady-state loop: a scratch spill or a re-materialised global load
would shift it, and the MFMAs would read a stage before it has landed - silently wrong moments.  This test compiles the file for
gfx950 and checks every basic block that holds both MFMAs and LDS-DMA loads."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "pybnesian_amd", "csrc")


@pytest.fixture(scope="module")
def stats_asm(tmp_path_factory):
    out = tmp_path_factory.mktemp("isa") / "stats_kernels.s"
    p = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-S", "--cuda-device-only",
                        "stats_kernels.hip", "-o", str(out)], cwd=CSRC, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    return out.read_text()


def kernels(asm, pattern):
    for f in re.split(r"\n(?=_Z[A-Za-z0-9_]+:)", asm):
        name = f.split(":", 1)[0]
        if re.search(pattern, name):
            yield name, f.split(".Lfunc_end")[0]   # the whole function: an early exit gives it more than one s_endpgm


def test_glds_ring_loops_hold_no_other_vector_memory_instruction(stats_asm):
    seen = 0
    for name, body in kernels(stats_asm, r"gram_glds(_f32)?_kernelILi[1-4]ELi0E"):
        seen += 1
        assert "scratch_" not in body, f"{name}: scratch spill"
        assert not re.search(r"\bbuffer_(load|store)", body), name
        blocks = re.split(r"\n(?=\.LBB\d+_\d+:)", body)
        ring = [b for b in blocks if "v_mfma_f64" in b and "global_load_lds" in b]
        assert ring, f"{name}: no steady-state block with MFMAs and LDS-DMA loads found"
        for b in ring:
            other = re.findall(r"^\s*(global_load_dword\w*|global_store\w*|flat_load\w*|flat_store\w*|global_atomic\w*)", b, flags=re.M)
            assert not other, f"{name}: {other[:3]} inside the ring loop shifts the hand-counted vmcnt"
    assert seen == 8   # NCT = 1..4, double and float tables


def test_gather_ring_kernels_keep_their_stages_in_flight(stats_asm):
    """gram_gring_kernel leaves its waits to the compiler, which counts them only for global_ loads in straight-line code: a flat_load
    (a pointer whose address space was lost) waits for everything, a scratch spill drains the queue, and a steady loop whose index
    wait is vmcnt(0) has one stage in flight instead of its ring."""
    seen = 0
    for name, body in kernels(stats_asm, r"gram_gring_kernelI[df]Li[1-4]ELb[01]E"):
        seen += 1
        assert "scratch_" not in body, f"{name}: scratch spill"
        assert "flat_load" not in body, f"{name}: flat load"
        blocks = re.split(r"\n(?=\.LBB\d+_\d+:)", body)
        steady = [b for b in blocks if "Inner Loop Header" in b and len(re.findall(r"v_mfma_f64", b)) >= 12 and "global_load_dword" in b]
        assert steady, f"{name}: no steady loop found"
        for b in steady[:1]:
            waits = [int(w) for w in re.findall(r"s_waitcnt vmcnt\((\d+)\)", b)]
            assert waits and min(waits) >= 4, f"{name}: the steady loop waits with vmcnt({min(waits) if waits else None})"
    assert seen == 16


@pytest.fixture(scope="module")
def kde_asm(tmp_path_factory):
    out = tmp_path_factory.mktemp("isa") / "kde_kernels.s"
    p = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-fno-slp-vectorize", "-S", "--cuda-device-only",
                        "kde_kernels.hip", "-o", str(out)], cwd=CSRC, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    return out.read_text()


def test_grouped_kernels_keep_their_register_budgets(kde_asm):
    """The grouped score-engine kernels live on their occupancy (profiles/r5/waves_probe.txt): the moment kernel of two-variable
    terms once ran 2x slower when a build spilled 42 of its coefficient registers, the fp64 sweeps are compiled for four waves per SIMD with two
    query groups per wave BECAUSE that shape fits 128 VGPRs nearly without scratch.  A compiler or source change that breaks this shows here,
    not as a silent slow-down."""
    found = {}
    for m in re.finditer(r"\.amdhsa_kernel (\S+)\n(.*?)\.end_amdhsa_kernel", kde_asm, flags=re.S):
        name, body = m.group(1), m.group(2)
        found[name] = (int(re.search(r"private_segment_fixed_size (\d+)", body).group(1)), int(re.search(r"next_free_vgpr (\d+)", body).group(1)))
    mom1, mom2 = found["_ZN3pbn23kde_moment_group_kernelILi1EEEvNS_10GSweepArgsE"], found["_ZN3pbn23kde_moment_group_kernelILi2EEEvNS_10GSweepArgsE"]
    assert mom1[0] == 0 and mom1[1] <= 168, mom1          # three waves per SIMD, nothing in scratch
    # round 6 (one tile walk for the wave's two query groups, the 16 queries of a group in one basic block): 45 coefficients per lane and still
    # <= 168 VGPRs - the 16 KB of per-(query, lane) sums in LDS, not the registers, set the occupancy (10 waves per CU)
    assert mom2[0] == 0 and mom2[1] <= 168, mom2
    for nm in ("_ZN3pbn23kde_moment_group_kernelILi1EEEvNS_10GSweepArgsE", "_ZN3pbn23kde_moment_group_kernelILi2EEEvNS_10GSweepArgsE"):
        body = re.search(r"\.amdhsa_kernel %s\n(.*?)\.end_amdhsa_kernel" % nm, kde_asm, flags=re.S).group(1)
        assert int(re.search(r"group_segment_fixed_size (\d+)", body).group(1)) == 16384, nm
    sweeps = {n: v for n, v in found.items() if n.startswith("_ZN3pbn22kde_sweep_group_kernelId")}
    assert len(sweeps) == 5, sorted(sweeps)               # KS = 1, 2 x norm folded / as weights, + the variant beside the moment pass
    for n, (scratch, vgpr) in sweeps.items():
        assert vgpr <= 128, (n, vgpr)                     # four waves per SIMD
        assert scratch <= 160, (n, scratch)               # (0-92 bytes today: the unit's argument block, no loop-carried spill)
    c5 = found["_ZN3pbn26kde_sweep_f16_group_kernelILi1EEEvNS_10GSweepArgsE"]
    assert c5[0] == 0 and c5[1] <= 128, c5                # C5's kernel (16x16 form): four waves per SIMD, nothing in scratch
    c5w = found["_ZN3pbn31kde_sweep_f16_w32p_group_kernelILi1EEEvNS_10GSweepArgsE"]
    assert c5w[0] == 0 and c5w[1] <= 128, c5w             # ... and its paired-tile 32x32x16 form (round 6): the same budget


def test_w32_sweep_stream_is_placed(kde_asm):
    """kde_sweep_f16_w32_kernel (the plain unpruned fp32 sweep): the steady loop is ONE basic block of four phases, and in every phase each
    v_mfma_f32_32x32x16_f16 is followed by its share of the previous accumulator's work - 16 / NJ v_exp_f32 and as many v_add_f32 - before the
    next MFMA issues (__builtin_amdgcn_sched_group_barrier in the source): no two MFMAs back to back, no exponential reading an accumulator
    younger than one phase (so no s_nop beyond the one slot the hazard recogniser wants at a phase boundary), nothing in scratch, and the
    register budgets of four (one 32-slot block) / three (two blocks) waves per SIMD."""
    seen = 0
    for nb, nj, vgpr_cap in ((1, 2, 128), (2, 4, 168)):
        m = re.search(r"\.amdhsa_kernel (_ZN3pbn24kde_sweep_f16_w32_kernelILi%dEEEvNS_9SweepArgsE)\n(.*?)\.end_amdhsa_kernel" % nb, kde_asm, flags=re.S)
        assert m, nb
        assert int(re.search(r"private_segment_fixed_size (\d+)", m.group(2)).group(1)) == 0
        assert int(re.search(r"next_free_vgpr (\d+)", m.group(2)).group(1)) <= vgpr_cap
        (name, body), = kernels(kde_asm, r"kde_sweep_f16_w32_kernelILi%dE" % nb)
        blocks = re.split(r"\n(?=\.LBB\d+_\d+:)", body)
        steady = [b for b in blocks if len(re.findall(r"v_mfma_f32_32x32x16_f16", b)) == 4 * nj and "Loop" in "\n".join(b.split("\n")[:3])]
        assert len(steady) == 1, (nb, len(steady))
        gaps = re.split(r"v_mfma_f32_32x32x16_f16[^\n]*\n", steady[0])[1:]
        for g in gaps:
            assert len(re.findall(r"\bv_exp_f32", g)) == 16 // nj, (nb, g)
            assert 16 // nj - 1 <= len(re.findall(r"\bv_add_f32", g)) <= 16 // nj + 2, (nb, g)
            assert all(int(x) <= 2 for x in re.findall(r"s_nop (\d+)", g)), (nb, g)
        assert "scratch_" not in steady[0]
        seen += 1
    assert seen == 2


def test_fp64_sum_only_sweeps_take_the_magic_form(kde_asm):
    """kde_sweep_kernel<double, KS, ..., EF32> (the C2 headline family, sum-only): 2^x comes from the accumulator's own words (exp2_magic) - the
    blind loops hold, per (tile, group) pair, four values x {v_alignbit_b32, v_exp_f32, v_cvt_f64_f32, v_lshl_add_u32, sum}: no v_fract_f64 /
    v_cvt_i32_f64 / v_cvt_f32_f64 / v_ldexp_f64 of a range reduction, the clamp (v_med3_i32) only in the bodies of tiles the guard did not
    prove, nothing in scratch, two waves per SIMD.  Two blind loops: chunks whose 32 tiles are all proven (two tiles per trip, no clamp
    anywhere), and chunks with an unproven tile (per tile a bare or a clamped body behind a scalar branch)."""
    seen = 0
    for ks, wmul in ((2, True), (1, True), (2, False), (3, False)):
        # kde_sweep_kernel<double, KS, COND=false, QG=4, FOLD=!wmul, PRUNE=false, WMUL, EF32=true>
        pat = r"kde_sweep_kernelIdLi%dELb0ELi4ELb%dELb0ELb%dELb1EE" % (ks, 0 if wmul else 1, 1 if wmul else 0)
        (name, body), = kernels(kde_asm, pat)
        hdr = re.search(r"\.amdhsa_kernel %s\n(.*?)\.end_amdhsa_kernel" % re.escape(name), kde_asm, flags=re.S).group(1)
        assert int(re.search(r"private_segment_fixed_size (\d+)", hdr).group(1)) == 0, name
        assert int(re.search(r"next_free_vgpr (\d+)", hdr).group(1)) <= 256, name
        blocks = re.split(r"\n(?=\.LBB\d+_\d+:)", body)

        def check(b, groups, clamped):
            for op in ("v_alignbit_b32", "v_exp_f32", "v_cvt_f64_f32", "v_lshl_add_u32"):
                assert len(re.findall(r"\b%s" % op, b)) == 4 * groups, (name, op)
            assert len(re.findall(r"\bv_med3_i32", b)) == (4 * groups if clamped else 0), name
            for op in ("v_fract_f64", "v_cvt_i32_f64", "v_cvt_f32_f64", "v_ldexp_f64", "scratch_"):
                assert op not in b, (name, op)
            sums = len(re.findall(r"\bv_fmac_f64|\bv_fma_f64", b)) if wmul else len(re.findall(r"\bv_add_f64", b))
            assert 3 * groups <= sums <= 5 * groups + 2, (name, sums)   # (the last add of a group may sit in the next block)

        exps = [b for b in blocks if "v_exp_f32" in b and "v_alignbit_b32" in b and "v_cmp_gt_f64" not in b]   # (not the checked loop's bodies)
        whole = [b for b in exps if len(re.findall(r"v_mfma_f64", b)) == 8 * ks]      # two tiles x four query groups per trip
        tiles = [b for b in exps if len(re.findall(r"v_mfma_f64", b)) == 4 * ks]      # one tile x four query groups
        assert len(whole) == 1 and len(tiles) == 4, (name, len(whole), len(tiles))
        check(whole[0], 8, False)
        clamps = sorted(len(re.findall(r"\bv_med3_i32", b)) for b in tiles)
        assert clamps == [0, 0, 16, 16], (name, clamps)
        for b in tiles:
            check(b, 4, "v_med3_i32" in b)
        seen += 1
    assert seen == 4


def test_pruned_sum_only_sweeps_keep_their_masks_scalar(kde_asm):
    """kde_sweep_group_kernel<double, KS = 1, QG = 2, norm in a K slot> (cv64's and C3's workhorse) and its stand-alone twin: the blind loop of
    proven batches takes exp2_magic without the clamp, the checked loop with it - and the 64-bit visit masks of the walk stay in SCALAR
    registers.  Round 6 measured what happens otherwise: a per-group gate kept as wave state across the walk pushed the masks into vector
    registers, every (tile, group) test became v_and_b32 / v_cmp_ne_u64 / s_and_saveexec, scratch went from 12 to 144 bytes and the kernel lost
    25 % - with every test green."""
    seen = 0
    for pat in (r"kde_sweep_group_kernelIdLi1ELi2ELb1ELb0ELb0EE", r"kde_sweep_kernelIdLi1ELb0ELi2ELb1ELb1ELb0ELb1EE"):
        (name, body), = kernels(kde_asm, pat)
        blocks = re.split(r"\n(?=\.LBB\d+_\d+:)", body)
        tails = [b for b in blocks if "v_exp_f32" in b and "v_alignbit_b32" in b]
        assert len(tails) >= 8, (name, len(tails))
        bare = [b for b in tails if "v_med3_i32" not in b]
        clamped = [b for b in tails if "v_med3_i32" in b]
        assert len(bare) >= 4 and len(clamped) >= 4, (name, len(bare), len(clamped))
        for b in tails:
            assert "v_cmp_ne_u64" not in b and "scratch_" not in b, name          # masks tested with s_and_b64 / s_cmp / s_bitcmp, nothing spilled here
        seen += 1
    assert seen == 2
