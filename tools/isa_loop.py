#!/usr/bin/env python3
"""Instruction mix of the sweep loop of the strip / fold kernels, from a device assembly file (hipcc -S --cuda-device-only).
usage: tools/isa_loop.py file.s [substring of the mangled kernel name ...]
The sweep loop is taken to be the smallest loop that holds >= 20 v_pk_fma_f32."""
import re
import sys
from collections import Counter


def functions(lines):
    out, name, start = {}, None, None
    for i, l in enumerate(lines):
        m = re.match(r'^(_Z\w+):', l)
        if m:
            name, start = m.group(1), i
        elif l.startswith('.Lfunc_end') and name:
            out[name] = lines[start:i]
            name = None
    return out


def sweep_loop(f):
    labels = {}
    for i, l in enumerate(f):
        m = re.match(r'^(\.LBB\d+_\d+):', l)
        if m:
            labels[m.group(1)] = i
    best = None
    for i, l in enumerate(f):
        m = re.search(r's_c?branch\w*\s+(\.LBB\d+_\d+)', l)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            body = [x.strip() for x in f[labels[m.group(1)]:i + 1] if x.strip() and not x.strip().startswith((';', '.'))]
            if sum(1 for x in body if x.startswith('v_pk_fma')) >= 20 and (best is None or len(body) < len(best)):
                best = body
    return best


def classify(body):
    c = Counter()
    for l in body:
        op = l.split()[0]
        code = l.split(';')[0]
        if op.startswith('v_pk_'):
            k = 'v_pk'
        elif 'dpp' in code or 'wave_sh' in code or 'row_' in code:
            k = 'dpp'
        elif op.startswith(('v_mov', 'v_accvgpr')):
            k = 'v_mov'
        elif op.startswith('v_'):
            k = 'v_other'
        elif op.startswith(('ds_', 'scratch_', 'global_', 'buffer_')):
            k = op
        elif op.startswith(('s_waitcnt', 's_barrier', 's_nop')):
            k = op
        elif op.startswith(('s_cbranch', 's_branch')):
            k = 'branch'
        elif op.startswith('s_'):
            k = 's_other'
        else:
            k = op
        c[k] += 1
    return c


def main():
    lines = open(sys.argv[1]).read().split('\n')
    pats = sys.argv[2:]
    for name, f in functions(lines).items():
        if pats and not any(p in name for p in pats):
            continue
        body = sweep_loop(f)
        if not body:
            continue
        c = classify(body)
        valu_cycles = 4.2 * (c['v_pk'] + c['dpp']) + 2.3 * (c['v_mov'] + c['v_other'])
        print('%s\n  loop: %d instr, issue estimate %.0f cycles per wavefront-sweep  %s' % (name, len(body), valu_cycles, dict(sorted(c.items()))))
        if '--dump' in sys.argv:
            print('\n'.join(body))


if __name__ == '__main__':
    main()
