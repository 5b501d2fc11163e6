import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
pk=d["roofline"]["per_kernel"]
print("ms/step %.3f"%d["ms_per_step"], {k:round(pk[k]["avg_launch_us"],1) for k in ("delta_cols","delta_rows","flow_y_fwd","x_grad","dphi_reduce") if k in pk})
