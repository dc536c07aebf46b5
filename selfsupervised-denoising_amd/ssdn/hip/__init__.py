"""MI355X (gfx950) backend of the ssdn hot path: ctypes binding of libssdn_hip.so, the op-list planner and the engine."""
