#!/usr/bin/env python3
"""Instruction mix of the sweep loop of the strip / fold kernels, from a device assembly file
(hipcc -S --cuda-device-only ...).   usage: tools/isa_loop.py file.s [substring of the mangled kernel name ...] [--dump] [--inner]
The sweep loop = the strongly connected component of the kernel's control-flow graph that holds the most
v_pk_fma_f32 (every block counted once, i.e. all rows active).  The issue estimate prices packed and DPP
instructions at 4.2 cycles and the other VALU instructions at 2.3 (profiles/r01_ubench_valu.txt)."""
import re
import sys
from collections import Counter

sys.setrecursionlimit(100000)


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


def blocks(f):
    """[(label, [instructions])] in layout order; successors by label."""
    bl, cur, name = [], [], 'entry'
    for l in f[1:]:
        m = re.match(r'^(\.LBB\d+_\d+):', l)
        t = l.strip()
        if m:
            bl.append((name, cur))
            name, cur = m.group(1), []
        elif t and not t.startswith((';', '.')):
            cur.append(t)
    bl.append((name, cur))
    succ = {}
    for i, (n, ins) in enumerate(bl):
        s = []
        fall = True
        for x in ins:
            m = re.match(r's_(c?)branch\w*\s+(\.LBB\d+_\d+)', x)
            if m:
                s.append(m.group(2))
                if not m.group(1):
                    fall = False
            if x.startswith('s_endpgm'):
                fall = False
        if fall and i + 1 < len(bl):
            s.append(bl[i + 1][0])
        succ[n] = s
    return bl, succ


def sccs(nodes, succ):
    index, low, on, st, out, c = {}, {}, set(), [], [], [0]

    def go(v):
        index[v] = low[v] = c[0]
        c[0] += 1
        st.append(v)
        on.add(v)
        for w in succ.get(v, []):
            if w not in index:
                go(w)
                low[v] = min(low[v], low[w])
            elif w in on:
                low[v] = min(low[v], index[w])
        if low[v] == index[v]:
            comp = []
            while True:
                w = st.pop()
                on.discard(w)
                comp.append(w)
                if w == v:
                    break
            out.append(comp)
    for v in nodes:
        if v not in index:
            go(v)
    return out


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
    pats = [a for a in sys.argv[2:] if not a.startswith('--')]
    for name, f in functions(lines).items():
        if pats and not any(p in name for p in pats):
            continue
        bl, succ = blocks(f)
        ins = dict(bl)
        best, bestn = None, 0
        for comp in sccs([n for n, _ in bl], succ):
            if len(comp) == 1 and comp[0] not in succ.get(comp[0], []):
                continue
            n = sum(1 for b in comp for x in ins[b] if x.startswith('v_pk_fma'))
            if n > bestn:
                best, bestn = comp, n
        if not best:
            continue
        if '--inner' in sys.argv:
            # persistent kernels: the phase loop contains the sweep loop.  Drop the blocks of the exchange (global
            # stores / loads, s_sleep) and take the component with the most packed multiply-adds of what is left.
            keep = [b for b in best if not any(x.startswith(('global_store', 'global_load', 'buffer_store', 'buffer_load', 's_sleep')) for x in ins[b])]
            sub = {b: [w for w in succ.get(b, []) if w in set(keep)] for b in keep}
            best2, bestn2 = None, 0
            for comp in sccs(keep, sub):
                if len(comp) == 1 and comp[0] not in sub.get(comp[0], []):
                    continue
                n = sum(1 for b in comp for x in ins[b] if x.startswith('v_pk_fma'))
                if n > bestn2:
                    best2, bestn2 = comp, n
            if best2:
                best = best2
        order = [n for n, _ in bl if n in set(best)]
        body = [x for n in order for x in ins[n]]
        c = classify(body)
        est = 4.2 * (c['v_pk'] + c['dpp']) + 2.3 * (c['v_mov'] + c['v_other'])
        print('%s\n  loop: %d blocks, %d instr, VALU issue estimate %.0f cycles per wavefront-sweep  %s' % (name, len(best), len(body), est, dict(sorted(c.items()))))
        if '--dump' in sys.argv:
            for n in order:
                print(n + ':')
                print('\n'.join('    ' + x for x in ins[n]))


if __name__ == '__main__':
    main()
