timeout 120 python3 tools/_nccl_probe.py 2>&1 | grep -E "avg supported|value|Error|error" | head
