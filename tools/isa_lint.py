#!/usr/bin/env python
"""ISA lint for the hazard of DESIGN.md section 4c: a VGPR that a DS instruction in flight still reads (address or data) is
overwritten by one of the next few instructions.  hipcc schedules such writes freely; on gfx950 the write raced the operand
read of the second of two back-to-back ds_bpermute when the writer was v_accvgpr_read_b32 (window-attention backward, round 3).

    hipcc --offload-arch=gfx950 -O3 -S --cuda-device-only -c kernel.hip -o kernel.s;  python tools/isa_lint.py kernel.s ...

Reports (kernel, line, DS instruction, overwriting instruction).  --all: any VALU writer, not only v_accvgpr_read_b32."""
import re
import sys


def regs(tok):
    tok = tok.strip().rstrip(',')
    m = re.fullmatch(r'v(\d+)', tok)
    if m:
        return {int(m.group(1))}
    m = re.fullmatch(r'v\[(\d+):(\d+)\]', tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    return set()


def main():
    args = [a for a in sys.argv[1:] if not a.startswith('--')]
    any_writer = '--all' in sys.argv
    window = 3
    total = 0
    for path in args:
        kern = None
        lines = open(path).read().split('\n')
        code = []
        for ln, line in enumerate(lines, 1):
            t = line.strip()
            if t.endswith(':') and not t.startswith('.') and not t.startswith(';'):
                kern = t[:-1]
            if not t or t.startswith(';') or t.startswith('.') or t.endswith(':'):
                continue
            code.append((ln, kern, t.split(';')[0].strip()))
        for i, (ln, k, ins) in enumerate(code):
            op = ins.split()[0]
            if not op.startswith('ds_'):
                continue
            ops = [o.strip() for o in ins[len(op):].split(',')]
            ops = [o.split()[0] for o in ops if o]
            if op.startswith('ds_read') or op.startswith('ds_load'):
                src = set().union(*[regs(o) for o in ops[1:2]])
            elif op.startswith('ds_bpermute') or op.startswith('ds_permute') or op.startswith('ds_swizzle'):
                src = set().union(*[regs(o) for o in ops[1:]])
            else:                                   # stores: address + data
                src = set().union(*[regs(o) for o in ops])
            if not src:
                continue
            for ln2, k2, ins2 in code[i + 1:i + 1 + window]:
                op2 = ins2.split()[0]
                if op2.startswith('s_') and op2 not in ('s_nop',):
                    if op2.startswith('s_waitcnt') or op2.startswith('s_barrier') or op2.startswith('s_cbranch'):
                        break
                    continue
                if not op2.startswith('v_'):
                    continue
                if not any_writer and op2 != 'v_accvgpr_read_b32':
                    continue
                if op2.startswith('v_cmp') or op2.startswith('v_mfma') or op2.startswith('v_accvgpr_write'):
                    continue
                dst = regs(ins2[len(op2):].split(',')[0].strip().split()[0])
                if dst & src:
                    total += 1
                    print(f'{path.split("/")[-1]}:{ln} [{(k or "")[:60]}] {ins}   <-   {ins2} (line {ln2})')
                    break
    print(f'{total} site(s)')


if __name__ == '__main__':
    main()
