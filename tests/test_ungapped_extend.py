"""The no-DP rule of the seed-extension kernel (mpibwa_amd/csrc/c2a_kernel.hip: `ungapped`), restated on the CPU
(tests/csrc/ungapped_extend_check.c) and compared with the oracle's ksw_extend2 restatement on seeded random flanks —
low-complexity ones, several scoring schemes, windows shorter than the flank, ambiguous bases: whenever the rule applies,
all six outputs must be the DP's."""
import os
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


@pytest.fixture(scope="module")
def checker(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("uec") / "ungapped_extend_check")
    subprocess.run(["gcc", "-O2", "-o", exe, os.path.join(HERE, "csrc", "ungapped_extend_check.c"), os.path.join(ROOT, "oracle", "orc_ksw.c"), "-lm"],
                   check=True)
    return exe


@pytest.mark.parametrize("seed", [21, 22, 23])
def test_closed_form_equals_the_dp(checker, seed):
    r = subprocess.run([checker, "150000", str(seed)], capture_output=True, text=True)
    cases, closed, bad = (int(x) for x in r.stdout.split())
    assert r.returncode == 0 and bad == 0, r.stderr
    assert closed > cases // 5   # the rule is not vacuous
