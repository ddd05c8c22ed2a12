for w in 20 16; do
LGMI_PERM_WPC=$w LGMI_LIB=$PWD/l-giremi_amd/build_abl/liblgmi_w5.so timeout -k 10 200 python bench.py --no-cpu-baseline --no-host-to-host --steps 2 --warmup 1 > gpurun_out/w5_$w.json 2> gpurun_out/w5_$w.err
python -c "import json; d=json.load(open('gpurun_out/w5_$w.json')); print('w5 lib WPC $w perm_general ms', round(d['stage_ms']['ms_perm_general'],1))"
done
