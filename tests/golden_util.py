import gzip
import json
import os

import numpy as np

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def kernel_vectors():
    return np.load(os.path.join(G, "kernel_vectors.npz"))


def ragged(kv, name):
    flat, off = kv[name], kv[name + "_off"]
    return [flat[off[i]:off[i + 1]] for i in range(len(off) - 1)]


def load_reads(name):
    out = []
    with gzip.open(os.path.join(G, name), "rt") as f:
        for line in f:
            n, a, b = line.rstrip("\n").split("\t")
            out.append((n, a.encode(), b.encode() if b else None))
    return out


def sam_cases():
    return json.load(open(os.path.join(G, "sam_cases.json")))


def load_sam(case):
    return gzip.open(os.path.join(G, "sam_%s.txt.gz" % case), "rb").read()


def golden_index(tmpdir):
    """Unpack the committed genome and index it with the product's builder."""
    from mpibwa_amd import api
    fa = os.path.join(str(tmpdir), "gold.fa")
    with gzip.open(os.path.join(G, "genome.fa.gz"), "rb") as g, open(fa, "wb") as f:
        f.write(g.read())
    api.build_index(fa, fa)
    return fa
