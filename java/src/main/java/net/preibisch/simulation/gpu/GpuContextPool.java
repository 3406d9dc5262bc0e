package net.preibisch.simulation.gpu;

/**
 * One native context per Java thread (mvsim contexts are bound to one GPU and are not thread-safe; the
 * reference calls convolve/extractSlices from two pool threads at once, SimulateTileStitching.java:85-117).
 * The device is taken from -Dmvsim.device=N (default 0).
 */
final class GpuContextPool
{
	private static final ThreadLocal< Long > CTX = ThreadLocal.withInitial( () -> {
		final long h = MvsimNative.create( Integer.getInteger( "mvsim.device", 0 ) );
		Runtime.getRuntime().addShutdownHook( new Thread( () -> MvsimNative.destroy( h ) ) );
		return h;
	} );

	private GpuContextPool() {}

	static long get() { return CTX.get(); }
}
