#!/usr/bin/env python3
"""scripts/isa_mix.py <file.s> <kernel-name-regex> [--loops]: instruction mix of one kernel of a hipcc -S listing
(whole kernel and per basic block with its label), to see where the VALU issue slots of a kernel go."""
import re, sys, subprocess, collections

def classify(op):
    if op.startswith('v_mfma') or op.startswith('v_smfmac'): return 'mfma'
    if op.startswith(('v_exp', 'v_rcp', 'v_rsq', 'v_sqrt', 'v_log', 'v_sin', 'v_cos')): return 'trans'
    if op.startswith('v_pk_'): return 'valu_pk'
    if op.startswith('v_cvt'): return 'valu_cvt'
    if op.startswith(('v_permlane', 'v_readlane', 'v_readfirstlane', 'v_writelane')) or '_dpp' in op: return 'valu_xlane'
    if op.startswith('v_'): return 'valu'
    if op.startswith('ds_'): return 'lds'
    if op.startswith(('global_', 'buffer_', 'flat_', 'scratch_')): return 'vmem'
    if op.startswith('s_waitcnt'): return 'wait'
    if op.startswith('s_barrier'): return 'barrier'
    if op.startswith('s_cbranch') or op.startswith('s_branch'): return 'branch'
    if op.startswith('s_load') or op.startswith('s_buffer_load'): return 'smem'
    if op.startswith('s_nop'): return 'nop'
    if op.startswith('s_'): return 'salu'
    return 'other'

def main():
    path, pat = sys.argv[1], sys.argv[2]
    show_blocks = '--loops' in sys.argv
    lines = open(path).read().split('\n')
    # kernel bodies: from "<mangled>:" to ".Lfunc_end"
    i = 0; found = []
    while i < len(lines):
        m = re.match(r'^(_Z\w+):\s*(;.*)?$', lines[i])
        if m:
            name = m.group(1)
            dem = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip()
            dem = re.sub(r'uf::\(anonymous namespace\)::', '', dem)
            j = i + 1
            while j < len(lines) and not lines[j].startswith('.Lfunc_end'): j += 1
            if re.search(pat, dem): found.append((dem, i + 1, j))
            i = j
        i += 1
    for dem, a, b in found:
        print('==', re.sub(r'\(.*', '', dem).replace('void ', ''))
        tot = collections.Counter(); blocks = []; cur = ['entry', collections.Counter(), 0]
        for l in lines[a:b]:
            l = l.split(';')[0].rstrip()
            if not l.strip(): continue
            m = re.match(r'^(\.LBB\w+):', l)
            if m:
                blocks.append(cur); cur = [m.group(1), collections.Counter(), 0]; continue
            if l.startswith('\t.') or l.startswith('.'): continue
            op = l.strip().split()[0]
            c = classify(op); tot[c] += 1; cur[1][c] += 1; cur[2] += 1
            if c == 'branch': cur.append(l.strip())
        blocks.append(cur)
        print('  total:', dict(sorted(tot.items())))
        if show_blocks:
            for blk in blocks:
                if blk[2] >= 24:
                    print(f'  {blk[0]:14s} n={blk[2]:5d}', dict(sorted(blk[1].items())), ' '.join(blk[3:])[:60])

if __name__ == '__main__':
    main()
