package net.preibisch.simulation.gpu;

import java.nio.FloatBuffer;
import java.util.Random;

import net.imglib2.IterableInterval;
import net.imglib2.RandomAccessibleInterval;
import net.imglib2.type.numeric.real.FloatType;
import net.imglib2.view.Views;

/** Drop-in for the numeric helpers of net.preibisch.simulation.Tools that sit on the per-view path (in place). */
public class ToolsGPU
{
	public static void poissonProcess( final RandomAccessibleInterval< FloatType > img, final double SNR, final Random rnd )
	{
		final FloatBuffer b = Buffers.toBuffer( Views.zeroMin( img ) );
		MvsimNative.poissonProcess( GpuContextPool.get(), b, b.capacity(), SNR, rnd.nextLong(), 0, 0L );
		Buffers.copyBack( b, Views.flatIterable( img ) );
	}

	public static void normImage( final Iterable< FloatType > img )
	{
		int n = 0;
		for ( @SuppressWarnings( "unused" ) final FloatType t : img ) ++n;
		final FloatBuffer b = Buffers.direct( n );
		for ( final FloatType t : img ) b.put( t.get() );
		b.rewind();
		MvsimNative.normImage( GpuContextPool.get(), b, n );
		Buffers.copyBack( b, img );
	}

	public static double adjustImage( final IterableInterval< FloatType > image, final float minValue, final float targetAverage )
	{
		final FloatBuffer b = Buffers.direct( image.size() );
		for ( final FloatType t : image ) b.put( t.get() );
		b.rewind();
		final double corr = MvsimNative.adjustImage( GpuContextPool.get(), b, image.size(), minValue, targetAverage );
		Buffers.copyBack( b, image );
		return corr;
	}
}
