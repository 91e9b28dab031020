for w in 32 64 128 256; do
python bench.py --windows $w --no-cpu-baseline --steps 20 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['roofline']['kernels']
print($w, round(d['value']), {n.replace('ba_','').replace('_kernel',''): round(v['ms_per_launch']*1e3,1) for n,v in k.items()})"
done
