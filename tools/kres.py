"""kernel resource usage from `make -C homulator_amd/csrc asm` (/tmp/hm_backend.resources): registers, scratch, occupancy, LDS per kernel.
usage: python tools/kres.py [substring ...]"""
import re, subprocess, sys
t = open('/tmp/hm_backend.resources').read()
pat = re.compile(r"Function Name: (\S+).*?\n.*?TotalSGPRs: (\d+).*?\n.*?VGPRs: (\d+).*?\n.*?AGPRs: (\d+).*?\n.*?ScratchSize \[bytes/lane\]: (\d+).*?\n.*?Dynamic Stack.*?\n.*?Occupancy \[waves/SIMD\]: (\d+).*?\n.*?SGPRs Spill: (\d+).*?\n.*?VGPRs Spill: (\d+).*?\n.*?LDS Size \[bytes/block\]: (\d+)")
rows = pat.findall(t)
names = subprocess.run(['c++filt'], input="\n".join(r[0] for r in rows), capture_output=True, text=True).stdout.split("\n")
for r, d in zip(rows, names):
    if len(sys.argv) > 1 and not any(s in d for s in sys.argv[1:]):
        continue
    print(f"{d[:78]:78s} sgpr {r[1]:>3s} vgpr {r[2]:>3s} scratch {r[4]:>4s} occ {r[5]} lds {r[8]}")
