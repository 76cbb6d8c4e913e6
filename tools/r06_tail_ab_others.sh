for m in ebu tp "ebu+tp --fs 44100" "ebu+tp --streams 1024 --seconds 60"; do for i in 1 2; do for t in 1 0; do
  echo -n "meters=$m tail=$t : "
  python bench.py --no-extra --no-cpu-baseline --steps 20 --warmup 5 --tail $t --meters $m 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); r=d['roofline']; print('k_ms %.4f gate_ms %.4f step_ms %.4f deferred %s' % (r['kernel_ms'], r['gate_ms'], d['ms_per_step'], d['config']['deferred_calls']))"
done; done; done
