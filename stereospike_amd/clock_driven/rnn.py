"""`spikingjelly.clock_driven.rnn` is imported (network/blocks.py:8) but never used by the reference; empty."""
