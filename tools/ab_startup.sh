for i in 1 2 3; do
echo "self 150:            $(PRE=self REPS=1 python tools/startup_latency.py 20 5 2>/dev/null | grep -E 'rep 0' | cut -c1-60)"
echo "clone 150:           $(PRE=train REPS=1 python tools/startup_latency.py 20 5 2>/dev/null | grep -E 'rep 0' | cut -c1-60)"
echo "self 150 + sleep 20: $(PRE=self SLEEP_MS=20 REPS=1 python tools/startup_latency.py 20 5 2>/dev/null | grep -E 'rep 0' | cut -c1-60)"
echo "self 150 + sleep 500:$(PRE=self SLEEP_MS=500 REPS=1 python tools/startup_latency.py 20 5 2>/dev/null | grep -E 'rep 0' | cut -c1-60)"
done
