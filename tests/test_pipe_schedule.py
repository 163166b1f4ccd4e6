"""The counted waits of gemm_huge.hip's free-running K loop (PIPE, round 5), restated and checked on the CPU.

The loop issues its LDS fragment reads ahead of the MFMAs that use them and waits with `s_waitcnt lgkmcnt(N)`: "at most N reads may
still be in flight".  LDS reads return in order, so the wait is right iff every fragment an MFMA uses was issued more than N
reads ago.  This test replays the program order of `pipe_step` (csrc/gemm_huge.hip: prologue W0 W1 A0..A3 W2; iteration j < 8:
wait, MFMA, read W[j + 3], 3 MFMAs (j = 0 waits per A fragment); iterations 8 + 9 merged: wait(1), then W'[1], A'[0..3], W'[2]
behind their last users) over several k-steps and checks (a) every operand has landed when its MFMA issues, (b) no read
overwrites a register an MFMA still needs (ring of four W registers, one set of A registers).  The constants below are the
ones in the source; the test also greps the source for them so that an edit of one side fails."""
import os
import re

SRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "world-in-world_amd", "csrc", "gemm_huge.hip")

WAIT_J0 = (4, 4, 3, 2)      # lgkmcnt before MFMA 0..3 of iteration 0 (W[0] + A[0], A[1], A[2], A[3])
WAIT_J = 2                  # iterations 1..7
WAIT_89 = 1                 # merged iterations 8 and 9
AHEAD = 3                   # W fragments are read three ahead
RING = 4


class Lds:
    def __init__(self):
        self.issued, self.landed = [], set()

    def read(self, name):
        self.issued.append(name)

    def wait(self, n):
        for x in self.issued[: max(len(self.issued) - n, 0)]:
            self.landed.add(x)

    def need(self, *names):
        for x in names:
            assert x in self.landed, f"{x} used before it landed (in flight: {self.issued[-8:]})"


def w(step, idx):            # fragment idx of k-step `step` (idx >= 10: of the next k-step)
    return f"W{step + idx // 10}_{idx % 10}"


def a(step, mi):
    return f"A{step}_{mi}"


def test_counted_waits_cover_every_operand():
    lds = Lds()
    ring = {}                                   # register slot -> fragment it holds (or will hold)
    last_use = {}                               # fragment -> True once its last MFMA has issued

    def read_w(step, idx):
        g = 10 * step + idx
        slot = g % RING
        old = ring.get(slot)
        assert old is None or last_use.get(old), f"read of {w(step, idx)} overwrites {old} before its last MFMA"
        ring[slot] = w(step, idx)
        lds.read(w(step, idx))

    for name in (w(0, 0), w(0, 1)):
        lds.read(name)
    ring[0], ring[1] = w(0, 0), w(0, 1)
    for mi in range(4):
        lds.read(a(0, mi))
    lds.read(w(0, 2))
    ring[2] = w(0, 2)
    for s in range(7):
        for j in range(8):
            if j == 0:
                for mi in range(4):
                    lds.wait(WAIT_J0[mi])
                    lds.need(w(s, 0), a(s, mi))
                    if mi == 0:
                        read_w(s, AHEAD)
            else:
                lds.wait(WAIT_J)
                lds.need(w(s, j), *(a(s, mi) for mi in range(4)))
                if j == 7 and s % 2 == 1:
                    lds.wait(0)                 # the tile's barrier: everything has landed
                read_w(s, j + AHEAD)
            last_use[w(s, j)] = True
        lds.wait(WAIT_89)
        lds.need(w(s, 8), w(s, 9))
        read_w(s, 11)                           # behind MFMA(8, 0): the registers of W[7]
        for mi in range(4):
            lds.read(a(s + 1, mi))              # behind MFMA(8, mi), MFMA(9, mi)
        last_use[w(s, 8)] = last_use[w(s, 9)] = True
        read_w(s, 12)                           # behind the last MFMA: the registers of W[8]


def test_source_holds_these_constants():
    src = open(SRC).read()
    body = src[src.index("auto pipe_step = [&]"):src.index("for (int kt = 0; kt < nk; ++kt) {", src.index("auto pipe_step = [&]"))]
    assert re.findall(r'lgkmcnt\((\d)\)" : "\+v"\(w_now\), "\+v"\(pa\[0\]\)', body) == [str(WAIT_J0[0])]
    assert [int(x) for x in re.findall(r'lgkmcnt\((\d)\)" : "\+v"\(pa\[[123]\]\)', body)] == list(WAIT_J0[1:])
    assert "else HP_WAIT(2, w_now);" in body and WAIT_J == 2
    assert re.search(r'lgkmcnt\(1\)" : "\+v"\(w8\), "\+v"\(w9\)', body) and WAIT_89 == 1
    assert "pb[(2 * KK + j + 3) & 3]" in body and "pb[(2 * KK + 11) & 3]" in body and "pb[(2 * KK + 12) & 3]" in body
    assert AHEAD == 3 and RING == 4
