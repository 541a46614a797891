import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import subprocess, re, sys
from cloud_transformers_amd import _lib
_lib.HIPCC_FLAGS.append("-Rpass-analysis=kernel-resource-usage"); _lib.HIPCC_FLAGS.extend(os.environ.get("CT_EXTRA", "").split())
import io, contextlib
try:
    r = subprocess.run([_lib._hipcc()] + _lib.HIPCC_FLAGS + ["-I", _lib.INCLUDE, _lib.CSRC + "/ct_raster.hip", "-o", "/tmp/raster_only.so"], capture_output=True, text=True)
except Exception as e:
    print(e); sys.exit(1)
txt = r.stderr
if r.returncode: print(txt[-3000:]); sys.exit(1)
cur=None; rows={}
for line in txt.splitlines():
    m=re.search(r'Function Name: (\S+)', line)
    if m: cur=m.group(1); rows[cur]={}
    for k in ('VGPRs:', 'VGPRs Spill:', 'SGPRs Spill:', 'ScratchSize [bytes/lane]:', 'Occupancy [waves/SIMD]:'):
        if k in line and cur:
            rows[cur][k]=line.split(k)[1].split()[0]
pat = sys.argv[1] if len(sys.argv)>1 else 'fused|hot|gather_ci'
for k,v in rows.items():
    if re.search(pat,k):
        print(subprocess.run(['c++filt',k],capture_output=True,text=True).stdout.strip()[:90], v)
