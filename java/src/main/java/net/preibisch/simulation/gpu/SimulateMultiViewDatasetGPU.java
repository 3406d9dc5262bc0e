package net.preibisch.simulation.gpu;

import java.nio.FloatBuffer;
import java.util.Random;
import java.util.concurrent.ExecutorService;

import mpicbg.models.AffineModel3D;
import net.imglib2.Interval;
import net.imglib2.RandomAccessibleInterval;
import net.imglib2.img.Img;
import net.imglib2.img.array.ArrayImgFactory;
import net.imglib2.type.numeric.real.FloatType;
import net.imglib2.view.Views;

/**
 * Drop-in for the per-view operators of net.preibisch.simulation.SimulateMultiViewDataset: identical static
 * signatures and return types, arithmetic on an MI355X through libmvsim.  Call sites
 * (SimulateMultiViewDataset.main :557-593, SimulateTileStitching :93-96, :108-111, :152-180) compile unchanged
 * after an import swap.
 *
 * Random numbers: the reference consumes one sequential java.util.Random stream inside the Poisson loop; here
 * ONE nextLong() is drawn from the caller's Random per call and keys a counter-based generator, so callers that
 * pass new Random(seed) stay reproducible (distributional, not stream, parity: DESIGN.md).
 */
public class SimulateMultiViewDatasetGPU
{
	final static Random rnd = new Random( 464232194 );
	final public static float minValue = 0.0001f;
	public final static float avgIntensity = 1;

	public static AffineModel3D axisRotation( final Interval in, final int axis, final int degrees )
	{
		final double[] m = new double[ 12 ];
		MvsimNative.axisRotation( Buffers.dims( in ), axis, degrees, m );
		final AffineModel3D model = new AffineModel3D();
		model.set( m[ 0 ], m[ 1 ], m[ 2 ], m[ 3 ], m[ 4 ], m[ 5 ], m[ 6 ], m[ 7 ], m[ 8 ], m[ 9 ], m[ 10 ], m[ 11 ] );
		return model;
	}

	public static Img< FloatType > rotateAroundAxis( final RandomAccessibleInterval< FloatType > in, final int axis, final int degrees )
	{
		final long[] d = Buffers.dims( in );
		final FloatBuffer out = Buffers.direct( Buffers.size( d ) );
		MvsimNative.rotateAroundAxis( GpuContextPool.get(), Buffers.toBuffer( Views.zeroMin( in ) ), d, axis, degrees, out );
		return Buffers.toImg( out, d );
	}

	public static Img< FloatType > attenuate3d( final RandomAccessibleInterval< FloatType > in, final double delta )
	{
		final long[] d = Buffers.dims( in );
		final FloatBuffer out = Buffers.direct( Buffers.size( d ) );
		MvsimNative.attenuate3d( GpuContextPool.get(), Buffers.toBuffer( Views.zeroMin( in ) ), d, delta, out );
		return Buffers.toImg( out, d );
	}

	/** The ExecutorService is accepted for signature compatibility and ignored. psf is normalised in place. */
	public static Img< FloatType > convolve( final Img< FloatType > img, final Img< FloatType > psf, final ExecutorService service )
	{
		final long[] d = Buffers.dims( img ), k = Buffers.dims( psf );
		final FloatBuffer p = Buffers.toBuffer( psf );
		final FloatBuffer out = Buffers.direct( Buffers.size( d ) );
		MvsimNative.convolve( GpuContextPool.get(), Buffers.toBuffer( img ), d, p, k, 0, out );
		Buffers.copyBack( p, psf );   // Tools.normImage( psf ) side effect of the reference (:255)
		return Buffers.toImg( out, d );
	}

	public static Img< FloatType > extractSlices( final RandomAccessibleInterval< FloatType > in, final int inc, final float poissonSNR )
	{
		return extractSlices( in, inc, poissonSNR, rnd );
	}

	public static Img< FloatType > extractSlices( final RandomAccessibleInterval< FloatType > in, final int inc, final float poissonSNR, final Random rnd )
	{
		final long[] d = Buffers.dims( in );
		final long[] o = new long[] { d[ 0 ], d[ 1 ], ( d[ 2 ] - 1 ) / inc + 1 };
		final FloatBuffer out = Buffers.direct( Buffers.size( o ) );
		final long seed = poissonSNR >= 0.0 ? rnd.nextLong() : 0L;
		MvsimNative.extractSlices( GpuContextPool.get(), Buffers.toBuffer( Views.zeroMin( in ) ), d, inc, poissonSNR, seed, 0, out );
		return Buffers.toImg( out, o );
	}

	public static Img< FloatType > poissonProcess( final RandomAccessibleInterval< FloatType > in, final float poissonSNR, final Random rnd )
	{
		final long[] d = Buffers.dims( in );
		final FloatBuffer b = Buffers.toBuffer( Views.zeroMin( in ) );
		MvsimNative.poissonProcess( GpuContextPool.get(), b, Buffers.size( d ), poissonSNR, rnd.nextLong(), 0, 0L );
		final long[] shape = new long[ in.numDimensions() ];
		in.dimensions( shape );
		return Buffers.toImg( b, in.numDimensions() == 3 ? d : new long[] { d[ 0 ], d[ 1 ], 1 } );
	}

	public static Img< FloatType > makeIsotropic( final RandomAccessibleInterval< FloatType > in, final int inc )
	{
		final long[] d = Buffers.dims( in );
		final long[] o = new long[] { d[ 0 ], d[ 1 ], ( d[ 2 ] - 1 ) * inc + 1 };
		final FloatBuffer out = Buffers.direct( Buffers.size( o ) );
		MvsimNative.makeIsotropic( GpuContextPool.get(), Buffers.toBuffer( Views.zeroMin( in ) ), d, inc, out );
		return Buffers.toImg( out, o );
	}

	public static Img< FloatType > computeWeightImage( final RandomAccessibleInterval< FloatType > in, final double delta )
	{
		final long[] d = Buffers.dims( in );
		final FloatBuffer out = Buffers.direct( Buffers.size( d ) );
		MvsimNative.computeWeightImage( GpuContextPool.get(), d, out );
		return Buffers.toImg( out, d );
	}

	/** simulate (:366-392): sphere cloud at 2x resolution, then 2x down-sampling; rnd must be a {@link GpuRandom}. */
	public static Img< FloatType > simulate( final boolean halfPixelOffset, final Random rnd )
	{
		final int scale = 2;
		final long n = ( 289 + 1 ) * scale;
		Img< FloatType > img = new ArrayImgFactory< FloatType >().create( new long[] { n, n, n }, new FloatType() );
		drawSpheres( img, 0, 1, scale, halfPixelOffset, rnd );
		return downSample2x( img );
	}

	/** drawSpheres (:436-522), in place: the walk over the large sphere on the host, the small spheres on the GPU. */
	public static void drawSpheres( final Img< FloatType > img, final double minValue, final double maxValue, final int scale,
			final boolean halfPixelOffset, final Random rnd )
	{
		if ( !( rnd instanceof GpuRandom ) )
			throw new IllegalArgumentException( "drawSpheres on the GPU continues the caller's random stream: pass a GpuRandom" );
		final GpuRandom g = ( GpuRandom ) rnd;
		final long[] state = new long[] { g.getState() };
		final FloatBuffer b = Buffers.toBuffer( img );
		MvsimNative.drawSpheres( GpuContextPool.get(), b, Buffers.dims( img ), minValue, maxValue, scale, halfPixelOffset, state );
		g.setState( state[ 0 ] );
		Buffers.copyBack( b, img );
	}

	/** downSample2x (:394-424) */
	public static Img< FloatType > downSample2x( final RandomAccessibleInterval< FloatType > in )
	{
		final long[] d = Buffers.dims( in );
		final long[] o = new long[] { d[ 0 ] / 2 - 1, d[ 1 ] / 2 - 1, d[ 2 ] / 2 - 1 };
		final FloatBuffer out = Buffers.direct( Buffers.size( o ) );
		MvsimNative.downSample2x( GpuContextPool.get(), Buffers.toBuffer( Views.zeroMin( in ) ), d, out );
		return Buffers.toImg( out, o );
	}

	/**
	 * The block of main() after the view loop (:615-640): per voxel, sum the weights of all views; zero sum gives
	 * zero weights, otherwise w = min( 1, osem * w / sum ).  In place on the given images.
	 */
	public static void normalizeWeights( final java.util.List< Img< FloatType > > weights, final float osem )
	{
		final FloatBuffer[] b = new FloatBuffer[ weights.size() ];
		for ( int i = 0; i < b.length; ++i )
			b[ i ] = Buffers.toBuffer( weights.get( i ) );
		MvsimNative.normalizeWeights( GpuContextPool.get(), b, b[ 0 ].capacity(), osem );
		for ( int i = 0; i < b.length; ++i )
			Buffers.copyBack( b[ i ], weights.get( i ) );
	}

	/** One view of the main loop (:570-585) with intermediates kept in HBM; returns { rot, att, con, acq }. */
	public static Img< FloatType >[] simulateView( final RandomAccessibleInterval< FloatType > groundTruth, final Img< FloatType > psf,
			final int degrees, final double attenuation, final int lightsheetSpacing, final float poissonSNR, final Random rnd, final int view )
	{
		final long[] d = Buffers.dims( groundTruth ), k = Buffers.dims( psf );
		final long[] o = new long[] { d[ 0 ], d[ 1 ], ( d[ 2 ] - 1 ) / lightsheetSpacing + 1 };
		final FloatBuffer p = Buffers.toBuffer( psf );
		final FloatBuffer rot = Buffers.direct( Buffers.size( d ) ), att = Buffers.direct( Buffers.size( d ) ),
				con = Buffers.direct( Buffers.size( d ) ), acq = Buffers.direct( Buffers.size( o ) );
		MvsimNative.simulateView( GpuContextPool.get(), Buffers.toBuffer( Views.zeroMin( groundTruth ) ), d, p, k, 0, degrees, attenuation,
				minValue, avgIntensity, lightsheetSpacing, poissonSNR, rnd.nextLong(), view, rot, att, con, acq );
		Buffers.copyBack( p, psf );
		@SuppressWarnings( "unchecked" )
		final Img< FloatType >[] res = new Img[] { Buffers.toImg( rot, d ), Buffers.toImg( att, d ), Buffers.toImg( con, d ), Buffers.toImg( acq, o ) };
		return res;
	}
}
