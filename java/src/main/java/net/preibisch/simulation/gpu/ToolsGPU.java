package net.preibisch.simulation.gpu;

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
		try ( Buffers.Block b = Buffers.toBlock( Views.zeroMin( img ) ) )
		{
			MvsimNative.poissonProcess( GpuContextPool.get(), b.floats, Buffers.size( Buffers.dims( img ) ), SNR, rnd.nextLong(), 0, 0L );
			Buffers.copyBack( b, Views.flatIterable( img ) );
		}
	}

	public static void normImage( final Iterable< FloatType > img )
	{
		long n = 0;
		for ( @SuppressWarnings( "unused" ) final FloatType t : img ) ++n;
		try ( Buffers.Block b = Buffers.direct( n ) )
		{
			for ( final FloatType t : img ) b.floats.put( t.get() );
			b.floats.rewind();
			MvsimNative.normImage( GpuContextPool.get(), b.floats, n );
			Buffers.copyBack( b, img );
		}
	}

	public static double adjustImage( final IterableInterval< FloatType > image, final float minValue, final float targetAverage )
	{
		try ( Buffers.Block b = Buffers.direct( image.size() ) )
		{
			for ( final FloatType t : image ) b.floats.put( t.get() );
			b.floats.rewind();
			final double corr = MvsimNative.adjustImage( GpuContextPool.get(), b.floats, image.size(), minValue, targetAverage );
			Buffers.copyBack( b, image );
			return corr;
		}
	}
}
