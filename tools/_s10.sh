cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -i -E "^\s*(Name|name)?.*(HBM|UMC|DRAM|MALL|EA_|TCC_EA|BUBBLE)" | head -60
echo ----
rocprofv3 -L 2>/dev/null | grep -c .
rocprofv3 -L 2>/dev/null | grep -i -o -E "\b(TCC_[A-Z0-9_]*(MALL|DRAM|EA0_RD|EA0_WR|IO)[A-Z0-9_]*)\b" | sort -u | head -60
