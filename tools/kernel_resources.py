#!/usr/bin/env python
"""Diagnostic: registers / scratch / LDS / occupancy of every kernel of one .hip file, as hipcc's -Rpass-analysis=kernel-resource-usage reports them
(no GPU needed).  usage: kernel_resources.py lra_amd/csrc/sdp.hip [more.hip ...]"""
import os
import re
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off"]


def demangle(names):
    try:
        out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
        return [re.sub(r"\(anonymous namespace\)::", "", x) for x in out]
    except OSError:
        return names


def main(paths):
    for p in paths:
        with tempfile.TemporaryDirectory() as d:
            r = subprocess.run(["hipcc", *FLAGS, "-Rpass-analysis=kernel-resource-usage", "-c", p, "-o", os.path.join(d, "x.o")], capture_output=True, text=True)
        ks, cur = [], None
        for l in r.stderr.split("\n"):
            m = re.search(r"remark:\s+Function Name: (\S+)", l)
            if m:
                cur = {"name": m.group(1)}; ks.append(cur); continue
            m = re.search(r"remark:\s+(SGPRs Spill|VGPRs Spill|TotalSGPRs|VGPRs|AGPRs|ScratchSize|Occupancy|LDS Size)[^:]*: (\d+)", l)
            if m and cur is not None:
                cur[m.group(1)] = int(m.group(2))
        names = demangle([k["name"] for k in ks])
        print("# %s" % p)
        print("%-64s %5s %5s %5s %7s %4s %7s %6s %6s" % ("kernel", "VGPR", "AGPR", "SGPR", "scratch", "occ", "LDS", "vspill", "sspill"))
        for k, n in zip(ks, names):
            if "rocprim" in n or "hipcub" in n:
                continue
            print("%-64s %5d %5d %5d %7d %4d %7d %6d %6d" % (n[:64], k.get("VGPRs", 0), k.get("AGPRs", 0), k.get("TotalSGPRs", 0), k.get("ScratchSize", 0),
                                                          k.get("Occupancy", 0), k.get("LDS Size", 0), k.get("VGPRs Spill", 0), k.get("SGPRs Spill", 0)))


if __name__ == "__main__":
    main(sys.argv[1:])
