#!/usr/bin/env python3
"""Per-kernel SQ counters from a rocprofv3 --pmc pass (rocpd .db): matrix-pipe and issue utilisation evidence.

    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY \
              SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_F32 SQ_INSTS_VALU --kernel-trace -d out -o sq -- python tools/pmc_probe.py
    python tools/pmc_sq_summary.py out/.../sq_results.db

Units (MI355X_MICROARCH.md): SQ_VALU_MFMA_BUSY_CYCLES counts cycles, SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* quad-cycles.
Values are summed over the device per launch (rocprofv3 reports the accumulated value of a dispatch)."""
import re
import sqlite3
import sys


def _mix_cycles():
    """Mean cycles per wave64 VALU instruction of the resident kernel's step loop (tools/isa_mix.py), 4.0 if none is committed.
    (It is the rollout kernel's mix; the other kernels' lines use it as the best available figure.)"""
    try:
        import json, os
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        return float(json.load(open(os.path.join(root, 'profiles', 'r06_isa_mix.json')))['mean_cycles_per_valu_instruction'])
    except Exception:
        return 4.0


def _source_hash():
    """Hash of the kernel sources the profiled library was built from (bench.py says whether the numbers are this build's)."""
    try:
        import os
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        from multiagent_gnn_policies_amd import build
        return build.source_hash()
    except Exception:
        return None


def main(path):
    import sys
    db = sqlite3.connect(path)
    rows = db.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection "
                      "group by kernel_name, counter_name").fetchall()
    dur = {}
    try:
        for name, n, avg in db.execute("select name, count(*), avg(duration) from kernels group by name"):
            dur[name] = (n, avg)
    except sqlite3.Error:
        pass
    per = {}
    for name, ctr, n, avg in rows:
        m = re.search(r'(\w+_kernel)', name)
        if not m or 'at::native' in name:
            continue
        per.setdefault(m.group(1), {})[ctr] = avg
        per[m.group(1)]['_n'] = n
        if name in dur:
            per[m.group(1)]['_us'] = dur[name][1] / 1e3
    ctrs = sorted({c for v in per.values() for c in v if not c.startswith('_')})
    print("# rocprofv3 --pmc (SQ block), average per launch, device totals")
    for k in sorted(per):
        v = per[k]
        print("%s  (launches %d%s)" % (k, v['_n'], ", %.1f us" % v['_us'] if '_us' in v else ''))
        for c in ctrs:
            if c in v:
                print("    %-28s %16.0f" % (c, v[c]))
        wc = v.get('SQ_WAVE_CYCLES')
        if wc:
            for c in ('SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_ANY'):
                if c in v:
                    print("    %-28s %15.1f%%  of wave cycles" % (c + ' share', 100.0 * v[c] / wc))
        if v.get('SQ_VALU_MFMA_BUSY_CYCLES') and v.get('SQ_BUSY_CU_CYCLES'):
            # one matrix pipe per SIMD, four SIMDs per CU: busy cycles are summed over the SIMDs
            print("    %-28s %15.1f%%  of (busy CU cycles x 4 SIMDs)" % ('matrix pipe busy', 100.0 * v['SQ_VALU_MFMA_BUSY_CYCLES'] /
                                                                        (4.0 * v['SQ_BUSY_CU_CYCLES'])))
        if v.get('SQ_INSTS_VALU') and v.get('SQ_BUSY_CU_CYCLES'):
            c = _mix_cycles()
            print("    %-28s %15.1f%%  of (busy CU cycles x 4 SIMDs), at %.2f cycles per wave instruction (%s)" %
                  ('vector pipes issuing', 100.0 * v['SQ_INSTS_VALU'] * c / (4.0 * v['SQ_BUSY_CU_CYCLES']), c,
                   'mix-weighted cost of the resident kernel\'s step loop: profiles/r06_isa_mix.json' if c != 4.0 else 'no ISA mix committed: flat 4'))


    if len(sys.argv) > 2:                                       # machine-readable fractions for bench.py's roofline.sq
        import json
        out = {}
        for k, v in per.items():
            wc = v.get('SQ_WAVE_CYCLES')
            if not wc:
                continue
            o = {"wait_any": v.get('SQ_WAIT_ANY', 0.0) / wc, "wait_inst_any": v.get('SQ_WAIT_INST_ANY', 0.0) / wc,
                 "active_inst_any": v.get('SQ_ACTIVE_INST_ANY', 0.0) / wc, "launches": v['_n'],
                 "avg_launch_us": v.get('_us'), "valu_insts": v.get('SQ_INSTS_VALU'), "mfma_f32_insts": v.get('SQ_INSTS_VALU_MFMA_F32')}
            if v.get('SQ_BUSY_CU_CYCLES'):
                o["mfma_busy"] = v.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0) / (4.0 * v['SQ_BUSY_CU_CYCLES'])
                # SQ_INSTS_VALU x (mix-weighted cycles per instruction, tools/isa_mix.py) / (4 SIMDs x busy CU cycles)
                o["valu_issue"] = v.get('SQ_INSTS_VALU', 0.0) * _mix_cycles() / (4.0 * v['SQ_BUSY_CU_CYCLES'])
                o["busy_cu_cycles"] = v['SQ_BUSY_CU_CYCLES']
            out[k] = o
        out['_meta'] = {"units": "fractions of SQ_WAVE_CYCLES (wave residency); mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / "
                                 "(4 SIMDs x SQ_BUSY_CU_CYCLES); valu_issue = SQ_INSTS_VALU x the mix-weighted cycles per instruction "
                                 "(tools/isa_mix.py) / (4 SIMDs x SQ_BUSY_CU_CYCLES)", "probe_T": __import__('os').environ.get('PROBE_T', '1000'),
                        "source_hash": _source_hash(),
                        # tools/pmc_probe.py: one 3-step launch, then five of PROBE_T steps, B = PROBE_B episodes: the average launch
                        # of the pass covers this many episode-steps (bench.py: VALU instructions per episode-step)
                        "rollout_episode_steps_per_launch": int(__import__('os').environ.get('PROBE_B', '256')) *
                        (3 + 5 * int(__import__('os').environ.get('PROBE_T', '1000'))) / 6.0}
        with open(sys.argv[2], 'w') as f:
            json.dump(out, f, indent=1)


if __name__ == '__main__':
    main(sys.argv[1])
