#!/bin/bash
# round 6, the committed build: the whole GPU suite, smoke(), the driver's bench
set -u
ROUND=6 bash tools/gpu_session.sh final tests_all smoke default
