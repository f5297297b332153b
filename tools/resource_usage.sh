#!/bin/bash
# usage: tools/resource_usage.sh [extra hipcc flags]  -> table of registers / scratch / occupancy of every kernel in libokenv
# (hipcc -Rpass-analysis=kernel-resource-usage on the shipped flags; commit the output under profiles/rN/)
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/_build
/opt/rocm/bin/hipcc -O3 -std=c++17 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt --offload-arch=gfx950 -fPIC -c \
  -Rpass-analysis=kernel-resource-usage "$@" -I include -I openkitchen_amd/csrc -o tools/_build/ru.o openkitchen_amd/csrc/okenv_capi.hip 2> tools/_build/ru.txt
python3 - <<'PY'
import re, subprocess
txt = open("tools/_build/ru.txt").read()
blocks = re.split(r"remark: [^\n]*Function Name: ", txt)[1:]
rows = []
for b in blocks:
    name = b.split("\n")[0].strip().split(" ")[0]
    g = lambda k: int(re.search(re.escape(k) + r": (\d+)", b).group(1)) if re.search(re.escape(k) + r": (\d+)", b) else -1
    rows.append((name, g("VGPRs"), g("AGPRs"), g("TotalSGPRs"), g("ScratchSize [bytes/lane]"), g("VGPRs Spill"), g("SGPRs Spill"), g("Occupancy [waves/SIMD]")))
names = subprocess.run(["c++filt"] + [r[0] for r in rows], capture_output=True, text=True).stdout.splitlines()
print("%-100s %5s %5s %5s %8s %7s %7s %4s" % ("kernel", "VGPR", "AGPR", "SGPR", "scratchB", "vspill", "sspill", "occ"))
for r, n in zip(rows, names):
    n = re.sub(r"\(OkStepParams.*", "", n).replace("void ", "")
    print("%-100s %5d %5d %5d %8d %7d %7d %4d" % ((n[:100],) + r[1:]))
PY
