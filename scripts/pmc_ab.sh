for so in raytracer_challenge_amd/csrc/variants/*.so; do
  RTC_AMD_LIB=$PWD/$so timeout -k 10 300 python3 bench.py --steps 50 --inflight 1 --no-cpu-baseline --extra-workloads "" 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); r = j['roofline']; t = r.get('traffic_detail', {})
        print('$(basename $so): seq %.3f ms  fetch %.2f GB write %.2f GB  l2hit %.2f  valu_busy %.2f lane_util %.2f wait %.2f' % (r['kernel_ms_avg'], t.get('fetch_bytes',0)/1e9, t.get('write_bytes',0)/1e9, t.get('l2_hit_rate',0), j['valu']['valu_busy_frac'], j['valu']['lane_utilisation'], j['valu']['wait_frac_of_wave_cycles']))
"
done
