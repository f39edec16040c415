#!/usr/bin/env python
"""ISA lint for the hand-scheduled GEMM kernels: the fragment reads are inline-asm ds_read_b128 whose results hipcc believes
to be ready at once.  Between such a read and the next `s_waitcnt lgkmcnt(0)` NO instruction may touch the destination
registers (a compiler-made copy would copy stale data; a late LDS return would clobber a re-used register).
Usage: python tools/check_async_frags.py <file.s> [kernel-name-substring]"""
import re, sys


def regs(tok):
    out = set()
    for m in re.finditer(r'\bv\[(\d+):(\d+)\]', tok):
        out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    for m in re.finditer(r'\bv(\d+)\b', tok):
        out.add(int(m.group(1)))
    return out


def main():
    s = open(sys.argv[1]).read()
    sub = sys.argv[2] if len(sys.argv) > 2 else ""
    bad = 0
    for m in re.finditer(r'^(\S+):\s*; @\1\n(.*?)\.end_amdhsa_kernel', s, re.S | re.M):
        name, body = m.group(1), m.group(2)
        if sub not in name:
            continue
        pending = set()
        nread = 0
        for n, line in enumerate(body.split('\n')):
            l0 = line.split(';')[0].strip()
            if not l0 or l0.startswith('.') or l0.endswith(':'):
                continue
            if l0.startswith('s_waitcnt') and 'lgkmcnt' in l0:
                pending.clear()
                continue
            if l0.startswith('ds_read_b128'):
                d = regs(l0.split(',')[0])
                a = regs(','.join(l0.split(',')[1:]))
                if a & pending:     # (a destination re-used by another read = dead result of the redundant tail reads: LDS returns in order)
                    print(f"{name}: line {n}: {l0}  touches pending {sorted(a & pending)}"); bad += 1
                pending |= d
                nread += 1
                continue
            t = regs(l0)
            if t & pending:
                print(f"{name}: line {n}: {l0}  touches pending {sorted(t & pending)[:8]}"); bad += 1
        print(f"{name[-60:]}: {nread} fragment reads checked")
    print("violations:", bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
