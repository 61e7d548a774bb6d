"""The tests' picture of which kernel serves which model (conftest.expected_member / wave_group_member) against the
library's own table (csrc/fsmc_capi.hip, w2Member; csrc/fsmc_instances.h, FSMC_ALL_W2) and the build's member list --
read from the sources, no GPU."""
import os
import re

from conftest import expected_member, wave_group_member

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "fastsmc_amd", "csrc")


def _w2_member_table():
    src = open(os.path.join(CSRC, "fsmc_capi.hip")).read()
    body = src[src.index("W2Member w2Member(int K)"):]
    body = body[:body.index("\n}\n")]
    steps = [(int(k), int(nw), int(kh)) for k, nw, kh in re.findall(r"if \(K <= (\d+)\) return \{(\d+), (\d+)\};", body)]
    last = re.search(r"\n  return \{(\d+), (\d+)\};", body)
    return steps, (int(last.group(1)), int(last.group(2)))


def test_wave_group_members_agree_with_the_library_and_the_build():
    steps, last = _w2_member_table()
    assert steps and steps == sorted(steps)
    kernels = open(os.path.join(CSRC, "fsmc_kernels.h")).read()
    k_max = int(re.search(r"constexpr int kMaxStatesW2 = (\d+);", kernels).group(1))
    built = set((int(nw), int(kh)) for kh, nw in
                re.findall(r"Y\((\d+), (\d+)\)", re.search(r"#define FSMC_ALL_W2\(Y\)(.*)", open(
                    os.path.join(CSRC, "fsmc_instances.h")).read()).group(1)))
    from fastsmc_amd.build import W2_MEMBERS
    assert built == set((nw, kh) for kh, nw in W2_MEMBERS)
    for K in range(129, k_max + 1):
        want = next(((nw, kh) for k, nw, kh in steps if K <= k), last)
        assert wave_group_member(K) == want, K
        assert want in built, (K, want)
        nw, kh = want
        # ghosts in the last wave (48-state member: the upper half; the members beyond 512 states: the last two waves)
        assert (nw - 1) * kh < K <= nw * kh or K <= 192 or (nw * kh > 512 and (nw - 2) * kh < K <= nw * kh), (K, want)
        assert expected_member(K) == (1000 + kh if nw == 4 else 1000 * nw + kh)
    assert expected_member(k_max + 1) == 0 and expected_member(69) == 69 and expected_member(70) == 80
