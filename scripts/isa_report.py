"""Developer tool: per-kernel instruction mix / scratch placement from a gfx950 assembly listing
(hipcc -O3 --offload-arch=gfx950 --cuda-device-only -S file.hip -o file.s).
usage: python scripts/isa_report.py file.s substring [substring ...]"""
import collections
import re
import sys


def is_ins(l):
    return l.startswith('\t') and l.strip() and not l.startswith('\t.') and not l.startswith('\t;')


def main():
    L = open(sys.argv[1]).read().split('\n')
    labels = [l[:-1].split(':')[0] for l in L if re.match(r'^_Z\w+:', l)]
    for nm in labels:
        if not all(k in nm for k in sys.argv[2:]):
            continue
        i = next(k for k, l in enumerate(L) if l.startswith(nm + ':'))
        j = i
        while not L[j].startswith('.Lfunc_end'):
            j += 1
        b = L[i:j]
        ins = [l.split()[0] for l in b if is_ins(l)]
        c = collections.Counter(ins)
        print(nm, 'instructions', len(ins))
        print('   top:', c.most_common(14))
        print('   scratch ops:', {k: v for k, v in c.items() if 'scratch' in k}, 'calls:', c.get('s_swappc_b64', 0))
        cur = None
        per = collections.Counter()
        size = collections.Counter()
        for l in b:
            if l.startswith('.LBB'):
                cur = l.split(':')[0]
            elif is_ins(l):
                size[cur] += 1
                if 'scratch' in l:
                    per[cur] += 1
        print('   blocks with scratch ops (block, scratch ops, block size):', [(k, v, size[k]) for k, v in per.items()])
        print('   largest blocks:', size.most_common(5))


main()
