"""profiles/r6_lockstep.txt from what tools/gpu_r6_lockstep.sh, gpu_r6_lockstep_trace.sh and gpu_r6_lockstep_shm.sh left under gpurun_out/."""
import json
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
O = os.path.join(ROOT, "gpurun_out")
L = ["Energy CG in lockstep with the velocity CG (round 6; DESIGN.md 6) - what one GPU can measure of it", ""]
L.append("1. One rank through the N-rank path (LGH_FORCE_MULTI=1; bench.py legs; tools/gpu_r6_lockstep.sh), two runs:")
for i in (1, 2):
    d = json.load(open(f"{O}/r6_lockstep/d_{i}.json"))
    L.append(f"   run {i}: headline {d['value']:.1f} Mdofs*steps/s, {d['ms_per_step']:.3f} ms per step")
    for n, g in d["legs"].items():
        L.append(f"     {n:13s} {g['value']:8.1f}  {g['ms_per_step']:7.3f} ms  (+{g['ms_per_step_minus_single_rank_path']:.3f} ms over the one-rank path)  lockstep {g.get('energy_lockstep')}")
L.append("   c2multi: second channel (energy beside velocity); c2multi1c: LGH_COMM2=0, lockstep; c2multi1cseq: LGH_COMM2=0 LGH_ENERGY_LOCKSTEP=0.")
L.append("   No peer, so an exchange costs nothing here: the lockstep's + 0.15-0.25 ms is what interleaving costs the energy kernels (2.)")
L.append("")
L.append("2. rocprofv3 --kernel-trace of 30 steps each (tools/gpu_r6_lockstep_trace.sh; LGH_FORCE_MULTI=1 LGH_COMM2=0):")
for mode, name in ((1, "lockstep"), (0, "energy after velocity")):
    t = open(f"{O}/r6_lockstep_trace/timeline_{mode}.txt").read().splitlines()
    L.append(f"   LGH_ENERGY_LOCKSTEP={mode} ({name}): " + [x for x in t if "RK4 steps" in x][0])
    for x in t:
        if "->" in x:
            continue
        if any(k in x for k in ("mass_apply_l2_kron", "l2_lockstep_fold_k", "vcg_apply_slab346", "vcg_update_p_k")):
            L.append("     " + x.strip())
L.append("   Between K1 and K2, which evict its vectors from the L2s, an energy kernel takes 1.7 us longer than back to back.")
L.append("")
L.append("3. Cross-process loop-back transport (ranks = processes sharing the GPU, every exchange a host round trip), 32^3 zones per rank,")
L.append("   `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N --transport shm --block 32 --steps 8 --warmup 3` (tools/gpu_r6_lockstep_shm.sh):")
for n in (2, 8):
    for tag, name in (("LGH_COMM2_0_LGH_ENERGY_LOCKSTEP_1", "one communicator, lockstep        "), ("LGH_COMM2_0_LGH_ENERGY_LOCKSTEP_0", "one communicator, energy after     "),
                      ("LGH_COMM2_1", "second channel (host threads fight)")):
        d = json.load(open(f"{O}/r6_lockstep_shm/d_{n}_{tag}.json"))
        L.append(f"   {n} ranks  {name}  {d['ms_per_step']:9.3f} ms per step  |e| = {d['config']['e_norm']!r}  t = {d['config']['t']!r}  {d['comm'].get('energy_lockstep')}")
L.append("   |e| and t agree to the last bit between the lockstep and the sequential order (the sums are formed in rank order either way).")
L.append("")
L.append("4. The lockstep energy kernels BESIDE K1 / K2 on the second stream (they exchange nothing, so no second communicator is needed), ordered")
L.append("   by four events per iteration (LGH_LOCKSTEP_STREAM2=1) - negative, kept as a switch with parity cases:")
for i in (1, 2):
    d = json.load(open(f"{O}/r6_lockstep/d1s_{i}.json"))
    g = d["legs"]["c2multi1c"]
    L.append(f"   run {i}: c2multi1c {g['value']:8.1f}  {g['ms_per_step']:7.3f} ms per step (+{g['ms_per_step_minus_single_rank_path']:.3f} ms over the one-rank path)")
L.append("   Two of the four waits sit on the velocity stream (before the messages are packed, before the word exchange): a cross-queue barrier")
L.append("   packet there costs more than the 11 us kernel it lets run beside K1 / K2.")
L.append("")
L.append("5. Tests (-m gpu): tests/test_gpu_pipeline.py::test_multi_rank_run_on_one_gpu[*lockstep*] (2, 4, 8 emulated ranks, problem 7, renumbered blocks,")
L.append("   both stream layouts; lgh_energy_lockstep_stats asserts the path ran), ::test_lockstep_energy_solve_is_bit_identical_to_the_sequential_order,")
L.append("   tests/test_gpu_multiproc.py[*one-communicator] (2 and 4 processes, bit for bit against the in-process ranks).")
open(os.path.join(ROOT, "profiles", "r6_lockstep.txt"), "w").write("\n".join(L) + "\n")
print("\n".join(L[:14]))
