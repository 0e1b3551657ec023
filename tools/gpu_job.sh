for cfg in "512 4" "640 4" "768 4" "768 6" "1024 4"; do set -- $cfg
python bench.py --plain --lanes $1 --engines $2 > gpurun_out/plain_l.json 2> gpurun_out/plain_l.err; python -c "
import json; r=json.load(open('gpurun_out/plain_l.json')); print('lanes $1 engines $2', r['value'], r['identical_to_reference'])" 2>&1 | tail -1
done
