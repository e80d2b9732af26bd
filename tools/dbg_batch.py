import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, R + '/tests'); sys.path.insert(0, R + '/oracle')
import numpy as np
import test_gpu_batch as t
g, reads, sets, make = t._paired_only(True)
one, many = make(), make()
want = [one.calc_prob(s) for s in sets]
got = many.calc_prob_batch(sets)
one.compact_tables(); many.compact_tables()
for rnd in range(2):
    want = [one.calc_prob(s) for s in sets]
    got = many.calc_prob_batch(sets)
print("classes", many.debug_class_counts(0), many.debug_table_stats(0))
b = [x[0] for x in many.calc_prob_batch(sets[:8])]
o = []
for i, s in enumerate(sets[:8]):
    v = many.calc_prob(s)[0]
    p1 = many.read_probs(0).copy()
    bb = many.calc_prob_batch([s])[0][0]
    p2 = many.read_probs(0).copy()
    print(i, repr(b[i]), repr(v), repr(bb), "EQ" if b[i] == v else "DIFF", "probs equal" if np.array_equal(p1, p2) else f"probs differ at {np.nonzero(p1 != p2)[0][:10]}")
print("single", repr(many.calc_prob(sets[4])[0]))
for combo in ([4], [4, 4], [4, 5], [5, 4], [3, 4], [0, 1, 2, 4], [4, 0, 1, 2], [0, 1, 2, 3, 4], [0, 1, 2, 3, 4, 5, 6, 7]):
    vals = many.calc_prob_batch([sets[k] for k in combo])
    print(combo, [repr(v[0]) for v, k in zip(vals, combo) if k == 4])
for knob in (3, 2):
    many.debug_set_knob(11, knob)
    vals = many.calc_prob_batch([sets[k] for k in range(8)])
    print("knob 11 =", knob, repr(vals[4][0]))
many.debug_set_knob(11, 0)
v1 = many.calc_prob(sets[4]); p1 = many.read_probs(0).copy()
v2 = many.calc_prob_batch([sets[4], sets[4]]); p2 = many.read_probs(0).copy()
print("single", repr(v1[0]), v1[1].tolist(), "batch", repr(v2[1][0]), v2[1][1].tolist(), "probs equal" if np.array_equal(p1, p2) else "probs differ")
# which classes matter: leave classes of blocks out of the multi kernel (timing knob: results wrong, but the difference tells)
for mask in (1, 2, 4, 8, 16):
    many.debug_set_knob(11, 32 + mask)
    a = many.calc_prob_batch([sets[4], sets[4]])[0][0]
    print("multi without class bit", mask, repr(a))
many.debug_set_knob(11, 0)
many.calc_prob(sets[4]); s1, z1, lay = many.debug_block_partials(0)
many.calc_prob_batch([sets[4], sets[4]]); s2, z2, lay2 = many.debug_block_partials(0, 0); s3, z3, _ = many.debug_block_partials(0, 1)
print("layout", lay, lay2, len(s1), len(s2))
d = np.nonzero(s1 != s2)[0]
print("blocks that differ single vs multi set 0:", d[:40], "of", len(s1))
for b in d[:10]:
    print(b, repr(s1[b]), repr(s2[b]), z1[b], z2[b])
