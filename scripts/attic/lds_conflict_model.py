"""Bank-conflict model of a tiled program's LDS gathers (device-free): per slot instruction of the fill rounds and of the usage pass, the
cycles a wave64 ds_read_b64 needs when the 64 lanes' addresses are served in passes of `group` lanes over 32 banks of 4 bytes (distinct
dwords on one bank serialise, equal addresses broadcast), against the conflict-free minimum.  usage: python scripts/lds_conflict_model.py <preset|c4b> [G] [K] [mode]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from machineboss_amd import capi
from machineboss_amd.machine import Machine
from machineboss_amd.evalmachine import EvaluatedMachine


def cycles(addr_bytes, group):
    """addr_bytes: 64 byte addresses of 8-byte reads."""
    tot = 0
    for g0 in range(0, 64, group):
        a = np.unique(addr_bytes[g0:g0 + group])
        dw = np.concatenate([a // 4, a // 4 + 1])
        banks = np.bincount((dw % 32).astype(np.int64), minlength=32)
        tot += int(banks.max())
    return tot


def main():
    name = sys.argv[1]; G = int(sys.argv[2]) if len(sys.argv) > 2 else 1; K = int(sys.argv[3]) if len(sys.argv) > 3 else 12
    mode = int(sys.argv[4]) if len(sys.argv) > 4 else 3
    if name == "c4b":
        from machineboss_amd import algebra
        m = algebra.config4bMachine("tests/golden/preset")
    else:
        m = Machine.fromFile("tests/golden/preset/%s.json" % name)
    em = EvaluatedMachine.fromMachine(m, None, useDefaults=True)
    prog = capi.debug_medium_program(em, "/tmp/prog.bin", mode=mode, closure=K, G=G)
    LPG, Spad, nOut = prog["LPG"], prog["Spad"], prog["nOut"]
    reps = 64 // LPG
    rng = np.random.RandomState(1)
    for group in (64, 32, 16):
        tot = ideal = n = 0
        for dp in prog["desc"]:
            ns = int(dp[0]) & 15
            for k in range(ns):
                for trial in range(4):      # columns of a wavefront hold their own tokens: average over a few random draws
                    addr = []
                    for c in range(reps):
                        it = rng.randint(1, prog["nIn"] + 1) if prog["nIn"] else 0; ot = rng.randint(1, min(nOut, 3) + 1) if nOut else 0
                        base = int(dp[1]) + it * int(dp[2]) + ot * (int(dp[3]) & 0xFFFFFF) + k * int(dp[4])
                        so = prog["rec"]["srcOff"][base:base + LPG].astype(np.int64) & 0xFFFF
                        addr.append(so + c * Spad * 8)
                    tot += cycles(np.concatenate(addr), group); ideal += 64 * 2 // 32; n += 1
        print("%s G=%d: fill-round gathers, passes of %d lanes: %.2f cycles per ds_read_b64 (conflict-free: %.1f), %d slot instructions" % (name, G, group, tot / n, ideal / n, n // 4))


if __name__ == "__main__":
    main()
