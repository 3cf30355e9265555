for s in 0 1 2 3 4 5; do echo "seed $s"; timeout -k 10 120 python examples/state_preparation.py 300 $s 2>&1 | grep -v Warning | tail -2; done
