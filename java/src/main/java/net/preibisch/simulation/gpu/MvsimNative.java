package net.preibisch.simulation.gpu;

import java.nio.FloatBuffer;

/**
 * JNI binding of libmvsim.so (C ABI in include/mvsim.h).  One native context per GPU; the context is
 * not thread-safe, so {@link GpuContextPool} hands one to each calling thread.
 *
 * SOURCE ONLY in this repository: the build image has no JDK (no javac, no jni.h); this layer has never been compiled
 * or run.  Compile on a host with a JDK: see INTEGRATION.md.
 */
final class MvsimNative
{
	static { System.loadLibrary( "mvsim_jni" ); }

	private MvsimNative() {}

	static native long create( int device );                          // mvsim_create
	static native void destroy( long ctx );                           // mvsim_destroy
	static native int deviceCount();                                  // mvsim_device_count

	// all buffers are DIRECT FloatBuffers (x fastest); dim = {nx, ny, nz}.  The shim checks every buffer's capacity against
	// the dimensions before it touches the C ABI and throws IllegalArgumentException when one is too small.
	static native void rotateAroundAxis( long ctx, FloatBuffer in, long[] dim, int axis, int degrees, FloatBuffer out );
	static native void attenuate3d( long ctx, FloatBuffer in, long[] dim, double delta, FloatBuffer out );
	static native void normImage( long ctx, FloatBuffer img, long n );
	static native void convolve( long ctx, FloatBuffer img, long[] dim, FloatBuffer psf, long[] kdim, int method, FloatBuffer out );
	static native double adjustImage( long ctx, FloatBuffer img, long n, float minValue, float targetAverage );
	static native void extractSlices( long ctx, FloatBuffer in, long[] dim, int inc, float snr, long seed, int stream, FloatBuffer out );
	static native void poissonProcess( long ctx, FloatBuffer img, long n, double snr, long seed, int stream, long indexOffset );
	static native void makeIsotropic( long ctx, FloatBuffer in, long[] dim, int inc, FloatBuffer out );
	static native void computeWeightImage( long ctx, long[] dim, FloatBuffer out );
	static native void axisRotation( long[] dim, int axis, int degrees, double[] m12 );

	// the per-stage operators with volumes as z-slab lists (mvsim_*_zslabs): in[i] holds inNz[i] planes, out[j] receives outNz[j];
	// this is how images beyond one 2 GiB direct buffer (2^29 voxels) cross the boundary -- a 512^3 image is a list of one
	static native void rotateAroundAxisSlabs( long ctx, FloatBuffer[] in, long[] inNz, long[] dim, int axis, int degrees, FloatBuffer[] out, long[] outNz );
	static native void attenuate3dSlabs( long ctx, FloatBuffer[] in, long[] inNz, long[] dim, double delta, FloatBuffer[] out, long[] outNz );
	static native void convolveSlabs( long ctx, FloatBuffer[] in, long[] inNz, long[] dim, FloatBuffer psf, long[] kdim, int method,
			FloatBuffer[] out, long[] outNz );
	static native void extractSlicesSlabs( long ctx, FloatBuffer[] in, long[] inNz, long[] dim, int inc, float snr, long seed, int stream,
			FloatBuffer[] out, long[] outNz );

	/** mvsim_host_alloc wrapped by NewDirectByteBuffer: a page-locked block (null if the allocation fails); freePinned releases it. */
	static native java.nio.ByteBuffer allocPinned( long ctx, long bytes );
	static native void freePinned( java.nio.ByteBuffer block );
	/**
	 * count floats between a direct FloatBuffer (from blockOffset) and a heap array (from arrayOffset) on the library's host threads
	 * (mvsim_host_copy; the array is held with GetPrimitiveArrayCritical meanwhile).  toArray: block -> array, else array -> block.
	 */
	static native void copyFloats( long ctx, java.nio.FloatBuffer block, long blockOffset, float[] array, int arrayOffset, int count, boolean toArray );

	/** drawSpheres (:436-522), in place; rndState[0] is the 48-bit java.util.Random state, advanced on return; returns the sphere count. */
	static native long drawSpheres( long ctx, FloatBuffer img, long[] dim, double minValue, double maxValue, int scale,
			boolean halfPixelOffset, long[] rndState );
	/** mvsim_splat_spheres: geometry = { cx, cy, cz, radius } per sphere, values = one float per sphere; in place. */
	static native void splatSpheres( long ctx, FloatBuffer img, long[] dim, int[] geometry, float[] values );
	/** downSample2x (:394-424): out has dim/2 - 1 samples per dimension. */
	static native void downSample2x( long ctx, FloatBuffer in, long[] dim, FloatBuffer out );

	/** Cross-view weight normalisation (:615-640), in place on every buffer. */
	static native void normalizeWeights( long ctx, FloatBuffer[] weights, long n, float osem );

	/** Fused loop body of SimulateMultiViewDataset.main (:570-585); rot/att/con may be null (then they never leave HBM). */
	static native double simulateView( long ctx, FloatBuffer gt, long[] dim, FloatBuffer psf, long[] kdim,
			int axis, int degrees, double delta, float minValue, float targetAverage, int inc, float snr, long seed, int stream,
			FloatBuffer rot, FloatBuffer att, FloatBuffer con, FloatBuffer acq );

	/** mvsim_simulate_view_async: returns a ticket at once; gt, psf and acq must stay reachable and untouched until waitView. */
	static native long simulateViewAsync( long ctx, FloatBuffer gt, long gtGeneration, long[] dim, FloatBuffer psf, long[] kdim,
			int axis, int degrees, double delta, float minValue, float targetAverage, int inc, float snr, long seed, int stream, FloatBuffer acq );
	/** mvsim_wait: blocks until the view has landed; returns the adjustImage factor. */
	static native double waitView( long ctx, long ticket );
	/** mvsim_wait without exceptions (cleanup paths). */
	static native void waitViewQuiet( long ctx, long ticket );

	// ---- one JVM process driving several GPUs: mvsim_group_* (GpuGroup) ----
	static native long groupCreate( int ndev );
	static native void groupDestroy( long group );
	static native void groupBroadcastVolume( long group, FloatBuffer gt, long[] dim );
	/** dim = the dimensions given to groupBroadcastVolume: the shim checks every acquisition buffer against them */
	/** mvsim_simulate_views: all views of one ground truth in ONE native call (stacked kernels for views that cannot fill the chip) */
	static native void simulateViewsBatch( long ctx, FloatBuffer gt, long[] dim, FloatBuffer[] psfs, long[] kdim, int[] degrees, double delta,
			float minValue, float targetAverage, int inc, float snr, long[] seeds, FloatBuffer[] acqs );

	static native void groupSimulateViews( long group, FloatBuffer[] psfs, long[] kdim, long[] dim, int[] degrees, double delta, float minValue,
			float targetAverage, int inc, float snr, long[] seeds, FloatBuffer[] acqs );
}
