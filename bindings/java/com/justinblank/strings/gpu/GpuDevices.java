package com.justinblank.strings.gpu;

/**
 * The GPUs of the node a batch is row-sharded over (needle_multi_*): one stream per device, and -- for more than one
 * distinct device -- an RCCL communicator for the gather of the results to devices[0].
 *
 * NOT COMPILED IN THE BUILD CONTAINER (no JDK); shipped as source.
 */
public final class GpuDevices implements AutoCloseable {
    private long handle;

    public GpuDevices(int... devices) {
        long[] h = new long[1];
        GpuPattern.check(Native.multiCreate(devices, 0, h), null);
        handle = h[0];
    }

    long handle() {
        return handle;
    }

    @Override
    public void close() {
        if (handle != 0) {
            Native.multiDestroy(handle);
            handle = 0;
        }
    }
}
