for v in "" "--variant hist_sort_launch=1" ""; do
python bench.py --plain $v > gpurun_out/plain_ab.json 2> gpurun_out/plain_ab.err; python -c "
import json; r=json.load(open('gpurun_out/plain_ab.json')); print('$v', r['value'], r['identical_to_reference'])"
done
