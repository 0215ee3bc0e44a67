set -e
python - <<'P'
import sys
sys.path.insert(0,'tests')
import streamgen
data,_=streamgen.write_stream(3840,2160,5,n_pictures=60,gop=2,bit_depth=10,wpp=1)
open('/tmp/s4k.bin','wb').write(data)
data,_=streamgen.write_stream(7680,4320,5,n_pictures=30,gop=2,bit_depth=10,wpp=1)
open('/tmp/s8k.bin','wb').write(data)
P
for f in /tmp/s4k.bin /tmp/s8k.bin; do
echo "== $f"
OHEVC_HOOK_TIMING=1 openhevc_amd/ohevc_dec -i $f -F oracle/_ref/libopenhevc_hip.so -c -n -p 16 -f 2 | tail -4
OHEVC_HOOK_TIMING=1 openhevc_amd/ohevc_dec -i $f -F oracle/_ref/libopenhevc_hip.so -c -n -g -p 16 -f 2 | tail -4
openhevc_amd/ohevc_dec -i $f -F oracle/_ref/libopenhevc_ref_sse.so -c -n -p 16 -f 2 | tail -1
done
