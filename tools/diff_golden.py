"""print the first SAM records that differ from the committed golden output (debugging aid)"""
import sys, tempfile
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
from golden_util import golden_index, load_reads, load_sam, sam_cases
from mpibwa_amd import api
with tempfile.TemporaryDirectory() as d:
    eng = api.Engine(golden_index(d), device=0)
    kw = sam_cases()["pe_default"]
    got = b"".join(eng.process(eng.opt(**kw), load_reads("reads_pe150.tsv.gz"))).split(b"\n")
    want = load_sam("pe_default").split(b"\n")
    k = 0
    for a, b in zip(got, want):
        if a != b:
            print("GOT ", a.decode()); print("WANT", b.decode()); k += 1
            if k >= 4: break
    print("differing shown:", k, "lines", len(got), len(want))
