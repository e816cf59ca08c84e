import sys,json
for l in sys.stdin:
    if l.startswith('{"metric'):
        j=json.loads(l); print(j["value"], j["ms_per_step"], "nn", j["roofline"]["kernel_ms_avg"], "red", j["roofline"]["reduce_ms_avg"], j["config"]["T_error_fro_vs_ground_truth"])
