#!/bin/bash
# Diagnostics (GPU box): poll power / clocks while a command runs.  usage: power_poll.sh <logfile> <cmd...>
LOG=$1; shift
( while true; do rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -E "sclk|Power \(W\)|junction" | tr '\n' ' '; echo; sleep 0.2; done ) > $LOG 2>&1 &
POLL=$!
"$@"
kill $POLL
