"""inputs of the -x tests (tests/test_host_cpu.py, tests/test_gpu_e2e.py): the same references and queries twice -- over a text alphabet
(what burst_hip gets) and with the symbols renamed, in the same order, to the bytes 1, 2, ... without 10 and 13 (what the compiled
reference gets: its query sort only survives symbols whose byte value is below 16, burst.c:383-387)"""
import numpy as np


def write_inputs(tmp_path, alphabet, seed, n_refs=40, n_queries=160):
    rng = np.random.default_rng(seed)
    sym = np.frombuffer("".join(sorted(alphabet)).encode(), np.uint8)
    low = np.array([b for b in range(1, 18) if b not in (10, 13)][:len(sym)], np.uint8)
    if low.max() > 15:
        return None
    refs = [rng.integers(0, len(sym), size=int(rng.integers(150, 420))) for _ in range(n_refs)]
    refs += [r.copy() for r in refs[:3]]                    # exact duplicates
    for r in refs[3:9]:                                     # near-identical variants: ties
        v = r.copy(); v[rng.integers(0, len(v), size=3)] = rng.integers(0, len(sym), size=3); refs.append(v)
    queries = []
    for i in range(n_queries):
        r = refs[int(rng.integers(len(refs)))]
        L = int(rng.choice([40, 60, 60, 90]))
        st = int(rng.integers(0, len(r) - L + 1)) if i % 9 else len(r) - L          # some reads end at the reference's last symbol
        q = r[st:st + L].copy()
        for _ in range(int(rng.integers(0, 4))):
            k = int(rng.integers(1, len(q) - 1)); t = int(rng.integers(3))
            if t == 0:
                q[k] = rng.integers(len(sym))
            elif t == 1:
                q = np.delete(q, k)
            else:
                q = np.insert(q, k, rng.integers(len(sym)))
        queries.append(q)

    def write(tag, table):
        rf, qf = str(tmp_path / (tag + "_r.fa")), str(tmp_path / (tag + "_q.fa"))
        with open(rf, "wb") as f:
            for i, r in enumerate(refs):
                f.write(b">ref%d\n" % i + table[r].tobytes() + b"\n")
        with open(qf, "wb") as f:
            for i, q in enumerate(queries):
                f.write(b">q%d\n" % i + table[q].tobytes() + b"\n")
        return rf, qf
    rf, qf = write("text", sym)
    rf_low, qf_low = write("low", low)
    return rf, qf, rf_low, qf_low
