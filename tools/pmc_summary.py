"""Condenses gpurun_out/prof/<tag>_pmc_hot.txt (tools/pmc_round.sh) into one line per kernel: where the waves' time
goes (waiting / issue-stalled / issuing), MFMA-pipe and VALU busy fractions, LDS activity and bank conflicts.
SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles, SQ_VALU_MFMA_BUSY_CYCLES cycles (MI355X_MICROARCH.md);
GRBM_GUI_ACTIVE is summed over the 8 XCDs.   usage: python tools/pmc_summary.py <file>"""
import collections
import re
import sys

txt = open(sys.argv[1]).read().split("\n")
data = collections.OrderedDict()
cur = None
for l in txt:
    if " grid=" in l and not l.startswith("    "):
        m = re.match(r"(.*?) grid=(\d+)", l)
        name = re.sub(r"void \(anonymous namespace\)::|\(anonymous namespace\)::", "", m.group(1))
        name = re.sub(r"\(gl_.*|\(anonymous.*", "", name)
        cur = (name, m.group(2))
        data.setdefault(cur, {})
    elif l.startswith("    ") and cur:
        p = l.split()
        data[cur][p[0]] = float(p[1])
print("%-46s %8s %7s | %6s %6s %6s | %6s %6s %7s %6s | %9s" % ("kernel", "threads", "us", "wait%", "stall%", "issue%", "MFMA%", "VALU%", "LDSact%", "bankcf", "VALU/MFMA"))
for (n, g), c in data.items():
    if "GRBM_GUI_ACTIVE" not in c:
        continue
    cyc = c["GRBM_GUI_ACTIVE"] / 8
    wc = c.get("SQ_WAVE_CYCLES", 1)
    print("%-46s %8s %7.1f | %6.1f %6.1f %6.1f | %6.1f %6.1f %7.1f %6.2f | %9.1f" % (
        n[:46], g, cyc / 2.1e3, 100 * c.get("SQ_WAIT_ANY", 0) / wc, 100 * c.get("SQ_WAIT_INST_ANY", 0) / wc, 100 * c.get("SQ_ACTIVE_INST_ANY", 0) / wc,
        100 * c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / 1024 / cyc, 100 * 4 * c.get("SQ_ACTIVE_INST_VALU", 0) / 1024 / cyc,
        100 * c.get("SQ_LDS_IDX_ACTIVE", 0) / 256 / cyc, c.get("SQ_LDS_BANK_CONFLICT", 0) / max(c.get("SQ_LDS_IDX_ACTIVE", 1), 1),
        c.get("SQ_INSTS_VALU", 0) / max(c.get("SQ_INSTS_MFMA", 1), 1)))
