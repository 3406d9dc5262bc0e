package net.preibisch.simulation.gpu;

/**
 * One native context per Java thread (mvsim contexts are bound to one GPU and are not thread-safe; the
 * reference calls convolve/extractSlices from two pool threads at once, SimulateTileStitching.java:85-117).
 * The device is taken from -Dmvsim.device=N (default 0).  {@link #release()} destroys the calling thread's context and
 * frees its pooled page-locked staging blocks; a shutdown hook does the same for threads that never call it.
 */
final class GpuContextPool
{
	private static final ThreadLocal< long[] > CTX = ThreadLocal.withInitial( () -> {
		final long[] h = new long[] { MvsimNative.create( Integer.getInteger( "mvsim.device", 0 ) ) };
		Runtime.getRuntime().addShutdownHook( new Thread( () -> {
			if ( h[ 0 ] != 0 )
				MvsimNative.destroy( h[ 0 ] );
			h[ 0 ] = 0;
		} ) );
		return h;
	} );

	private GpuContextPool() {}

	static long get()
	{
		final long[] h = CTX.get();
		if ( h[ 0 ] == 0 )
			h[ 0 ] = MvsimNative.create( Integer.getInteger( "mvsim.device", 0 ) );
		return h[ 0 ];
	}

	/** destroy the calling thread's context and give its page-locked staging memory back */
	static void release()
	{
		Buffers.drain();
		final long[] h = CTX.get();
		if ( h[ 0 ] != 0 )
			MvsimNative.destroy( h[ 0 ] );
		h[ 0 ] = 0;
	}
}
