import re, sys, collections
lines = open('/tmp/log.txt', errors='replace').read().splitlines()
pat = re.compile(r'tid: (0x[0-9a-f]+)\]')
out = []
kinds = collections.Counter()
for i, l in enumerate(lines):
    if 'hipMemcpyAsync (' in l and 'hipMemcpyDeviceToHost' in l and ', 16,' not in l:
        tid = pat.search(l).group(1)
        follow = []
        for m in lines[i + 1:i + 400]:
            if ('tid: ' + tid) in m:
                follow.append(m)
                if 'hipMemcpyAsync: Returned' in m: break
        key = ' | '.join(re.sub(r'0x[0-9a-f]+|\d+', 'N', f.split(']')[-1])[:90] for f in follow)
        kinds[key] += 1
        if len(out) < 3 or 'HSA Copy' not in key: out.append((l, follow))
for k, v in kinds.most_common(): print(v, k)
print()
for l, f in out[-3:]:
    print(l[:200]); [print('   ', x[:220]) for x in f]
