"""dev helper: per-basic-block instruction histogram of one kernel in a hipcc -S listing."""
import re, sys
from collections import Counter
s = open(sys.argv[1]).read()
name = [l[:-1] for l in re.findall(r'^_ZN5dgcnn\w+:', s, re.M) if sys.argv[2] in l][int(sys.argv[3]) if len(sys.argv) > 3 else 0]
print(name)
i = s.index('\n' + name + ':'); j = s.index('.Lfunc_end', i)
body = s[i:j].split('\n')
labels = [(n, l) for n, l in enumerate(body) if re.match(r'^\.LBB\d+_\d+:', l)]
labels = [(0, 'entry')] + labels
for a, (n, l) in enumerate(labels):
    e = labels[a + 1][0] if a + 1 < len(labels) else len(body)
    c = Counter()
    for x in body[n:e]:
        x = x.strip()
        if not x or x[0] in ';.' : continue
        c[x.split()[0]] += 1
    tot = sum(c.values())
    if tot >= int(sys.argv[4]) if len(sys.argv) > 4 else 20:
        print(l.split(';')[0].strip(), tot, sorted(c.items(), key=lambda kv: -kv[1])[:14])
