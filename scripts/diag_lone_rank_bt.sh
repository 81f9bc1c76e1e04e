#!/bin/bash
# Where is a lone rank of a 2-rank non-blocking communicator stuck?  (diagnosis of the RCCL 2.27.7 deadline problem)
# The box forbids attaching to a running process, so python runs under rocgdb from the start and is interrupted after 35 s.
mkdir -p gpurun_out/r5c
LOG=gpurun_out/r5c/lone_bt.log
/opt/rocm/bin/rocgdb -batch -ex "set pagination off" -ex "handle SIGINT stop nopass" -ex "handle SIG32 SIG33 SIG34 SIG35 nostop noprint pass" \
  -ex run -ex "thread apply all bt 40" -ex kill --args python scripts/diag_lone_rank.py > $LOG 2>&1 &
GDB=$!
sleep 40
CHILD=$(pgrep -P $GDB | head -1)
echo "gdb $GDB child $CHILD"
[ -n "$CHILD" ] && kill -INT $CHILD
for i in $(seq 1 60); do kill -0 $GDB 2>/dev/null || break; sleep 1; done
kill -9 $GDB 2>/dev/null
grep -n "Thread \|^#" $LOG | grep -v "alt_rsmi" | head -150
