"""Pick the LDS node strides (SI, SJ) of the 3-D stage-A cell image by counting bank conflicts
(MI355X_MICROARCH.md LDS table) for the lane->address maps of the three pencil directions and the
node-linear phases: ds_read_b64 = 32-lane groups over 32 double-banks; ds_write_b64 / ds_add_f64 =
16-lane groups over 16 double-banks."""
import sys
N = int(sys.argv[1]) if len(sys.argv) > 1 else 6

def cost(addrs, lanes, banks):
    extra = 0
    for g in range(0, len(addrs), lanes):
        grp = set(addrs[g:g + lanes])
        b = {}
        for a in grp:
            b.setdefault(a % banks, set()).add(a)
        if b:
            extra += max(len(v) for v in b.values()) - 1
    return extra

def evaluate(SI, SJ):
    SL = N * SI
    rd = wr = 0
    for d in range(3):
        tasks = [(l, t) for l in range(N) for t in range(N * N)]
        for w0 in range(0, len(tasks), 64):
            addrs = []
            for (l, t) in tasks[w0:w0 + 64]:
                a, b = divmod(t, N)
                addrs.append(l * SL + {0: a * SJ + b, 1: a * SI + b, 2: a * SI + b * SJ}[d])
            rd += cost(addrs, 32, 32)
            wr += cost(addrs, 16, 16)
    nodes = [i * SI + j * SJ + k for i in range(N) for j in range(N) for k in range(N)]
    for w0 in range(0, len(nodes), 64):
        rd += cost(nodes[w0:w0 + 64], 32, 32)
        wr += cost(nodes[w0:w0 + 64], 16, 16)
    return rd, wr

res = []
for SJ in range(N, N + 3):
    for SI in range(N * SJ, N * SJ + 9):
        rd, wr = evaluate(SI, SJ)
        res.append((rd + 2 * wr, rd, wr, SI * N, SI, SJ))     # a write/atomic pass costs ~2x a read pass
res.sort()
for r in res[:10]:
    print("score %3d (read %2d, write %2d extra passes)  slab %4d doubles  SI=%d SJ=%d" % r)
print("unpadded (SI=N*N, SJ=N):", evaluate(N * N, N))
