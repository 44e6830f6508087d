#!/usr/bin/env python3
"""Prices the vector-instruction mix of a kernel's step loop with MEASURED issue costs.

    python tools/isa_mix.py [--rates profiles/r06_valu_rate.txt] [--out profiles/r06_isa_mix.json]

What it does: compiles csrc/rollout.hip to gfx950 assembly (hipcc -S, the library's flags), takes the headline instantiation
rollout_kernel<100, 3, FD = 0, CL = 0, CM = 1, WBF = 1, VL = 1>, cuts out its step loop (the largest depth-1 loop), and sorts every
VALU instruction into the rate classes tools/harness/valu_rate.hip measured on MI355X (cycles a wave64 instruction occupies its
SIMD's vector pipe with four waves per SIMD issuing independent work):

    full    2.4   v_fma / v_fmac / v_mul / v_add / v_sub _f32, v_and / v_or / v_xor _b32, v_add / v_sub _u32, v_mov_b32
    half    4.3   everything else on the vector ALU: packed fp32, fp64, conversions, shifts, v_cndmask, v_max / v_min, integer
                  multiplies, v_lshl_add, v_perm, v_alignbit, v_bcnt, DPP forms of the full-rate instructions
    cmp     5.1   v_cmp_* (SGPR-pair result)      lane   5.6  v_readlane / v_readfirstlane / v_writelane
    trans   8.2   v_exp / v_rcp / v_rsq / v_sqrt / v_log _f32;  v_permlane*_swap 8.3;  fp64 transcendental 16.2

The static mix of the loop body stands in for the dynamic one (SQ_INSTS_VALU counts instructions, not classes): every
instruction once, whatever its wave count and trip count.  `bench.py` turns the mix-weighted mean cost c into the kernel's
vector-issue peak, CUs x 4 SIMDs x clock / c wave-instructions per second, and prices the measured SQ_INSTS_VALU rate with it.
"""
import argparse
import collections
import hashlib
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

FULL = {'v_fma_f32', 'v_fmac_f32', 'v_mul_f32', 'v_add_f32', 'v_sub_f32', 'v_subrev_f32', 'v_mac_f32', 'v_fmamk_f32', 'v_fmaak_f32',
        'v_and_b32', 'v_or_b32', 'v_xor_b32', 'v_not_b32', 'v_add_u32', 'v_sub_u32', 'v_subrev_u32', 'v_mov_b32', 'v_mul_legacy_f32'}
TRANS32 = {'v_exp_f32', 'v_rcp_f32', 'v_rsq_f32', 'v_sqrt_f32', 'v_log_f32', 'v_sin_f32', 'v_cos_f32', 'v_rcp_iflag_f32'}
TRANS64 = {'v_rcp_f64', 'v_rsq_f64', 'v_sqrt_f64'}
LANE = {'v_readlane_b32', 'v_readfirstlane_b32', 'v_writelane_b32'}
# fallbacks if a class is missing from the rates file (the numbers of profiles/r06_valu_rate.txt)
DEFAULT_RATES = {'full': 2.4, 'half': 4.3, 'cmp': 5.1, 'lane': 5.6, 'trans': 8.2, 'swap': 8.3, 'trans64': 16.2, 'mfma_bf16_16x16x32': 17.4}
RATE_ROWS = {'full': ['v_fma_f32', 'v_mul_f32', 'v_and_b32', 'v_add_u32', 'v_mov_b32'],
             'half': ['v_pk_fma_f32', 'v_fma_f64', 'v_mul_f64', 'v_add_f64', 'v_cvt_pk_bf16_f32', 'v_alignbit_b32', 'v_mov_b32 dpp row_shl:1',
                      'v_add_f32 dpp row_shr:1', 'v_cndmask_b32', 'v_mul_lo_u32', 'v_mad_u32_u24', 'v_bcnt_u32_b32', 'v_lshlrev_b64', 'v_perm_b32',
                      'v_and_or_b32', 'v_lshl_add_u32', 'v_lshlrev_b32', 'v_max_f32', 'v_cvt_f32_u32', 'v_cvt_f64_f32', 'v_cvt_f32_f64',
                      'v_mad_u64_u32'],
             'cmp': ['v_cmp_lt_f32 (sgpr pair)', 'v_cmp_lt_f64 (sgpr pair)'], 'lane': ['v_readlane_b32'],
             'trans': ['v_rcp_f32', 'v_exp_f32'], 'swap': ['v_permlane32_swap'], 'trans64': ['v_rcp_f64'],
             'mfma_bf16_16x16x32': ['v_mfma_f32_16x16x32_bf16']}


def load_rates(path):
    """class -> cycles per wave-instruction and SIMD at four waves per SIMD (third number of a row of the harness table)."""
    rates = dict(DEFAULT_RATES)
    if not path or not os.path.exists(path):
        return rates, 'built-in defaults (no rates file)'
    table = {}
    for line in open(path):
        if line.startswith('#') or '|' not in line:
            continue
        cols = [c.strip() for c in line.split('|')]
        try:
            table[cols[0]] = float(cols[1].split()[2])
        except (IndexError, ValueError):
            continue
    for cls, rows in RATE_ROWS.items():
        vals = [table[r] for r in rows if r in table]
        if vals:
            rates[cls] = sum(vals) / len(vals)
    return rates, os.path.relpath(path, ROOT)


def classify(mn, operands):
    base = re.sub(r'_(e32|e64|dpp|sdwa|e64_dpp)$', '', mn)
    dpp = mn.endswith('_dpp') or 'row_' in operands or 'quad_perm' in operands
    if base.startswith('v_mfma') or base.startswith('v_smfma'):
        return 'mfma'
    if base.startswith('v_permlane'):
        return 'swap'
    if base in LANE:
        return 'lane'
    if base.startswith('v_cmp') or base.startswith('v_cmpx'):
        return 'cmp'
    if base in TRANS32:
        return 'trans'
    if base in TRANS64:
        return 'trans64'
    if base in FULL and not dpp:
        return 'full'
    return 'half'


def step_loop(lines):
    """(first, last) line index of the largest depth-1 loop of the kernel body (LLVM's loop comments)."""
    headers = {}
    for i, ln in enumerate(lines):
        m = re.match(r'^(\.LBB\d+_\d+):.*Loop Header: Depth=1', ln)
        if m:
            headers[m.group(1)[2:]] = [i, i]
    for i, ln in enumerate(lines):
        m = re.search(r'Header=(BB\d+_\d+) Depth=1', ln)
        if m and m.group(1) in headers:
            headers[m.group(1)][1] = i
    # a block comment marks the START of a block: extend to the end of that last block (next label or s_endpgm)
    best = max(headers.values(), key=lambda v: v[1] - v[0])
    j = best[1] + 1
    while j < len(lines) and not re.match(r'^\.LBB\d+_\d+:', lines[j]) and 's_endpgm' not in lines[j]:
        j += 1
    return best[0], j


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--rates', default=os.path.join(ROOT, 'profiles', 'r06_valu_rate.txt'))
    ap.add_argument('--out', default=None)
    ap.add_argument('--asm', default=None, help='use this assembly file instead of compiling')
    ap.add_argument('--kernel', default=r'rollout_kernelILi100ELi3ELb0ELb0ELb1ELb1ELb1E')
    args = ap.parse_args()
    from multiagent_gnn_policies_amd import build as mb
    if args.asm:
        text = open(args.asm).read()
    else:
        with tempfile.TemporaryDirectory() as td:
            out = os.path.join(td, 'rollout.s')
            cmd = [mb.hipcc_path()] + mb.COMMON_FLAGS + mb.PER_FILE_FLAGS['rollout.hip'] + ['-w', '-S', '--cuda-device-only', '-o', out,
                                                                                            os.path.join(mb.CSRC, 'rollout.hip')]
            subprocess.run(cmd, check=True)
            text = open(out).read()
    lines = text.splitlines()
    start = next(i for i, ln in enumerate(lines) if re.match(r'^_Z\w*' + args.kernel + r'\w*:', ln))
    end = next(i for i in range(start, len(lines)) if 's_endpgm' in lines[i])
    body = lines[start:end + 1]
    lo, hi = step_loop(body)
    loop = body[lo:hi]
    rates, rates_src = load_rates(args.rates)
    counts = collections.Counter()
    by_mn = collections.Counter()
    other = collections.Counter()
    for ln in loop:
        ln = ln.split(';')[0].strip()
        if not ln or ln.startswith('.') or ln.endswith(':'):
            continue
        parts = ln.split(None, 1)
        mn, ops = parts[0], parts[1] if len(parts) > 1 else ''
        if mn.startswith('v_'):
            cls = classify(mn, ops)
            counts[cls] += 1
            by_mn[re.sub(r'_(e32|e64|dpp|sdwa)$', '', mn) + (' dpp' if (mn.endswith('_dpp') or 'row_' in ops or 'quad_perm' in ops) else '')] += 1
        else:
            other['lds' if mn.startswith('ds_') else 'vmem' if mn.startswith(('global_', 'buffer_', 'flat_', 'scratch_')) else
                  'salu' if mn.startswith('s_') else 'other'] += 1
    valu = sum(v for k, v in counts.items() if k != 'mfma')
    cyc = sum(rates[k] * v for k, v in counts.items() if k != 'mfma')
    res = {
        "kernel": "rollout_kernel<100, 3, FD=0, CL=0, CM=1, WBF=1, VL=1> (the headline instantiation), step loop",
        "source_hash": mb.source_hash(),
        "rates_file": rates_src,
        "cycles_per_wave_instruction": {k: round(v, 3) for k, v in rates.items()},
        "static_valu_instructions": valu,
        "by_class": {k: counts[k] for k in sorted(counts)},
        "mean_cycles_per_valu_instruction": cyc / valu,
        "mfma_instructions": counts.get('mfma', 0),
        "other_instructions": dict(other),
        "top_mnemonics": by_mn.most_common(24),
        "note": "static mix of the step loop body (every instruction once) priced with the four-waves-per-SIMD issue costs of "
                "tools/harness/valu_rate.hip; MFMA instructions run on the matrix pipe and are listed, not priced",
    }
    js = json.dumps(res, indent=1)
    if args.out:
        with open(args.out, 'w') as f:
            f.write(js + '\n')
    print(js)


if __name__ == '__main__':
    main()
