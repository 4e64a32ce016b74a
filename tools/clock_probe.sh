mkdir -p gpurun_out/clk
( timeout -k 10 120 python3 bench.py --gpus 1 --steps 30000 --warmup 50 --no-verify --no-host-io > gpurun_out/clk/bench.json 2> gpurun_out/clk/bench.err ) &
BP=$!
sleep 1
for i in $(seq 1 60); do
  rocm-smi --showclocks --showpower --showuse 2>/dev/null | grep -E "sclk|mclk|fclk|Power|GPU use" | tr '\n' ' ' >> gpurun_out/clk/smi.log
  echo >> gpurun_out/clk/smi.log
  sleep 0.5
  kill -0 $BP 2>/dev/null || break
done
wait $BP
echo rc=$?
tail -c 600 gpurun_out/clk/bench.json | head -c 400
echo
awk 'NR%4==0' gpurun_out/clk/smi.log | tail -12
