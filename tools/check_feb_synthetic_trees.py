#!/usr/bin/env python3
"""FindEquivalentBranches of this library against the reference binary on a synthetic tree sequence at any N
(no painting, no tree building: random binary trees, neighbours differing by `moves` subtree-prune-and-regraft moves
of single leaves; needs oracle/_ref/Relate; CPU only):

    python tools/check_feb_synthetic_trees.py [N trees moves seed]

N = 5000, 12 trees, 20 moves: reference 0.71 s, this library 0.25 s, same bytes."""
import os, shutil, struct, subprocess, sys, tempfile, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import rlutil

N, T, moves, seed = [int(x) for x in sys.argv[1:5]] if len(sys.argv) > 4 else (5000, 12, 20, 7)
rng = np.random.RandomState(seed)


def random_tree():
    active, nxt, left, right = list(range(N)), N, {}, {}
    while len(active) > 1:
        i, j = rng.choice(len(active), 2, replace=False)
        left[nxt], right[nxt] = active[i], active[j]
        for x in sorted((i, j), reverse=True):
            active.pop(x)
        active.append(nxt)
        nxt += 1
    return left, right, nxt - 1


def parents(left, right, root):  # internal nodes relabelled children first, the root last
    order, todo = [], [root]
    while todo:
        v = todo.pop()
        order.append(v)
        if v in left:
            todo += [left[v], right[v]]
    order = [v for v in reversed(order) if v in left]
    lab = {v: N + i for i, v in enumerate(order)}
    lab.update({l: l for l in range(N)})
    parent = np.full(2 * N - 1, -1, np.int32)
    for v in order:
        parent[lab[left[v]]] = parent[lab[right[v]]] = lab[v]
    return parent


def move_a_leaf(left, right, root):
    par = {}
    for v in left:
        par[left[v]] = par[right[v]] = v
    leaf = rng.randint(N)
    p = par[leaf]
    if p == root:
        return
    sib = right[p] if left[p] == leaf else left[p]
    g = par[p]
    if left[g] == p:
        left[g] = sib
    else:
        right[g] = sib
    par[sib] = g
    cand = [v for v in par if v not in (leaf, p)]
    t = cand[rng.randint(len(cand))]
    tp = par[t]
    if left[tp] == t:
        left[tp] = p
    else:
        right[tp] = p
    left[p], right[p] = leaf, t


work = tempfile.mkdtemp()
try:
    left, right, root = random_tree()
    cdir = os.path.join(work, "src", "out", "chunk_0")
    os.makedirs(cdir)
    W, per, k = 2, T // 2, 0
    open(os.path.join(work, "src", "out", "parameters_c0.bin"), "wb").write(struct.pack("<iii", N, 1000, W + 1))
    for w in range(W):
        with open(os.path.join(cdir, "out_%d.anc" % w), "wb") as f:
            f.write(struct.pack("<?II", False, N, per))
            for t in range(per):
                for m in range(moves):
                    move_a_leaf(left, right, root)
                f.write(struct.pack("<i", k))
                k += 1
                rec = np.zeros(2 * N - 1, dtype=[("p", "<i4"), ("bl", "<f8"), ("ne", "<f4"), ("sb", "<i4"), ("se", "<i4")])
                rec["p"], rec["bl"], rec["ne"], rec["sb"], rec["se"] = parents(left, right, root), 1.0, 1.0, k, k
                f.write(rec.tobytes())
    took = {}
    for who, exe in (("reference", rlutil.REF_RELATE), ("library", os.path.join(ROOT, "relate_amd", "Relate"))):
        shutil.copytree(os.path.join(work, "src"), os.path.join(work, who))
        t0 = time.time()
        subprocess.run([exe, "--mode", "FindEquivalentBranches", "--chunk_index", "0", "-o", "out"],
                       cwd=os.path.join(work, who), check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        took[who] = time.time() - t0
    same = all(open(os.path.join(work, "reference", "out", "chunk_0", "out_%d.anc" % w), "rb").read() ==
               open(os.path.join(work, "library", "out", "chunk_0", "out_%d.anc" % w), "rb").read() for w in range(W))
    print("N=%d, %d trees, %d moves between neighbours: reference %.2f s, library %.2f s, %s" %
          (N, 2 * per, moves, took["reference"], took["library"], "same bytes" if same else "DIFFERENT FILES"))
    sys.exit(0 if same else 1)
finally:
    shutil.rmtree(work, ignore_errors=True)
