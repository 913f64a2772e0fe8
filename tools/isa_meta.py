#!/usr/bin/env python
"""Register / spill metadata of every kernel in a gfx950 assembly file (hipcc -save-temps=obj ... *.s).

    python tools/isa_meta.py /tmp/isa/lf_conv-hip-amdgcn-amd-amdhsa-gfx950.s [name-substring]

Prints demangled name, VGPRs, AGPRs, spilled VGPRs / SGPRs, scratch bytes, LDS bytes.  Used by
tests/test_isa_cpu.py (no hot kernel may spill) and when tuning kernels without a GPU.
"""
import re
import subprocess
import sys


def kernels(asm_path):
    s = open(asm_path).read()
    md = s[s.index('amdhsa.kernels:'):]
    out = []
    for b in md.split('\n  - .agpr_count:')[1:]:
        def g(k, blk=b):
            m = re.search(r'\.' + k + r':\s+(\S+)', blk)
            return m.group(1) if m else None
        out.append(dict(mangled=g('name'), agpr=int(b.split('\n')[0].strip()), vgpr=int(g('vgpr_count')),
                        vgpr_spill=int(g('vgpr_spill_count')), sgpr_spill=int(g('sgpr_spill_count')),
                        scratch=int(g('private_segment_fixed_size')), lds=int(g('group_segment_fixed_size'))))
    # scratch accesses between the first and the last MFMA of each kernel's code: a spill INSIDE the matrix loop (the ones
    # before / behind it -- prologue, epilogue -- cost a dozen instructions per tile)
    body = {}
    cur = None
    for line in s[:s.index('amdhsa.kernels:')].split('\n'):
        if line and not line[0].isspace() and line.endswith(':') and not line.startswith('.'):
            cur = line[:-1]
            body[cur] = []
        elif cur is not None:
            body[cur].append(line)
    for k in out:
        lines = body.get(k['mangled'], [])
        mf = [i for i, l in enumerate(lines) if 'v_mfma' in l]
        k['loop_scratch'] = sum(1 for l in lines[mf[0]:mf[-1]] if 'scratch_' in l) if mf else 0
    names = subprocess.run(['c++filt'], input='\n'.join(k['mangled'] for k in out), capture_output=True, text=True).stdout.split('\n')
    for k, n in zip(out, names):
        n = n.replace('(anonymous namespace)::', '')
        k['name'] = re.sub(r'^void ', '', re.sub(r'\(.*', '', n))
    return out


if __name__ == '__main__':
    sub = sys.argv[2] if len(sys.argv) > 2 else ''
    for k in sorted(kernels(sys.argv[1]), key=lambda k: k['name']):
        if sub in k['name']:
            print('%-75s vgpr %3d agpr %3d spill v%3d s%3d scratch %5d (in the MFMA loop: %d) lds %6d' % (k['name'][:75], k['vgpr'], k['agpr'], k['vgpr_spill'], k['sgpr_spill'], k['scratch'], k['loop_scratch'], k['lds']))
