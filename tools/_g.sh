cd /tmp && export TMPDIR=/tmp
R=/root/repo
rocprofv3 --kernel-trace --output-format csv -d /tmp/pp -o p -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-alt > /tmp/pp.json 2>/tmp/pp.err
f=$(find /tmp/pp -name "*kernel_trace.csv" | head -1)
cp $f $R/gpurun_out/joint_trace.csv
python $R/tools/prof_seq.py $f 6.5 20 | grep -v "copyBuffer\|elementwise" | cut -c1-100
