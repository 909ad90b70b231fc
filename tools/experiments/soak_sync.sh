for i in 1 2 3 4 5 6 7 8 9 10; do
  DAZIM_FMM_ASYNC=0 python bench.py --steps 25 --warmup 0 --no-cpu 2>/dev/null | python /tmp/soak_line.py "sync run $i"
done
