"""Static view of one kernel's ISA: basic blocks with their instruction counts by class (which blocks are the straight-line hot
paths is obvious from their size). Usage: isa_blocks.py FILE.s MANGLED_SUBSTRING [min_instructions]"""
import sys, re, collections
path, key = sys.argv[1], sys.argv[2]
minn = int(sys.argv[3]) if len(sys.argv) > 3 else 40
lines = open(path).read().split('\n')
start = next(i for i, l in enumerate(lines) if l.startswith('_ZN') and key in l and l.rstrip().split(':')[0].endswith(key.split()[-1]) or (l.startswith('_ZN') and key in l.split(':')[0]))
end = next(i for i in range(start, len(lines)) if lines[i].startswith('.Lfunc_end'))
blocks, cur, name = [], [], 'entry'
for l in lines[start + 1:end]:
    if l.startswith('.LBB'):
        blocks.append((name, cur)); name, cur = l.split(':')[0], []
    elif l.startswith('\t') and not l.strip().startswith(('.', ';')):
        cur.append(l.strip())
blocks.append((name, cur))
FOUR = ('v_lshl', 'v_lshr', 'v_ashr', 'v_mad_i32_i24', 'v_mad_u32_u24', 'v_mul_i32_i24', 'v_mul_u32_u24', 'v_bfe', 'v_cmp', 'v_cvt', 'v_perm', 'v_alignb', 'v_pk_', 'v_dot2', 'v_frexp', 'v_mul_lo', 'v_mul_hi', 'v_mad_u64', 'v_lshl_add', 'v_lshl_or', 'v_and_or', 'v_or3', 'v_add3', 'v_bfi', 'v_sdwa')
tot = collections.Counter()
for name, ins in blocks:
    c = collections.Counter(i.split()[0] for i in ins)
    for k, n in c.items(): tot[k] += n
    if len(ins) < minn: continue
    v = sum(n for k, n in c.items() if k.startswith('v_'))
    v4 = sum(n for k, n in c.items() if k.startswith(FOUR))
    print(f"{name:12s} n={len(ins):5d} valu={v:5d} (4-cycle class ~{v4:4d}) ds={sum(n for k,n in c.items() if k.startswith('ds_')):3d} vmem={sum(n for k,n in c.items() if k.startswith(('global_','buffer_','flat_'))):3d} salu={sum(n for k,n in c.items() if k.startswith('s_')):4d}  top: " + ' '.join(f"{k}:{n}" for k, n in c.most_common(9)))
print('TOTAL', sum(tot.values()))
