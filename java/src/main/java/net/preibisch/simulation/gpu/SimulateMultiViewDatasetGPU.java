package net.preibisch.simulation.gpu;

import java.util.ArrayList;
import java.util.List;
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
 * pass new Random(seed) stay reproducible (distributional, not stream, parity: DESIGN.md).  drawSpheres, whose
 * stream the reference shares with everything else, consumes the caller's generator exactly as the reference does.
 *
 * SOURCE ONLY in this repository (no JDK in the build image): never compiled or run here (the JNI shim gets a syntax-only
 * compile against a hand-written jni.h subset, tests/jni_stub -- a compile check, it pins nothing).
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

	// The four per-stage operators take ANY image an ArrayImg can hold (up to 2^31 - 1 voxels, the reference's own limit,
	// SimulateMultiViewDataset.java:109): volumes cross the boundary as z-slab lists of page-locked blocks (Buffers.Slabs), one
	// block for everything up to 2^28 voxels, several beyond -- a 1024^3 image is four.

	public static Img< FloatType > rotateAroundAxis( final RandomAccessibleInterval< FloatType > in, final int axis, final int degrees )
	{
		final long[] d = Buffers.dims( in );
		try ( Buffers.Slabs src = Buffers.toSlabs( Views.zeroMin( in ) ); Buffers.Slabs out = new Buffers.Slabs( d[ 0 ], d[ 1 ], d[ 2 ] ) )
		{
			MvsimNative.rotateAroundAxisSlabs( GpuContextPool.get(), src.buffers(), src.nz, d, axis, degrees, out.buffers(), out.nz );
			return Buffers.toImg( out );
		}
	}

	public static Img< FloatType > attenuate3d( final RandomAccessibleInterval< FloatType > in, final double delta )
	{
		final long[] d = Buffers.dims( in );
		try ( Buffers.Slabs src = Buffers.toSlabs( Views.zeroMin( in ) ); Buffers.Slabs out = new Buffers.Slabs( d[ 0 ], d[ 1 ], d[ 2 ] ) )
		{
			MvsimNative.attenuate3dSlabs( GpuContextPool.get(), src.buffers(), src.nz, d, delta, out.buffers(), out.nz );
			return Buffers.toImg( out );
		}
	}

	/** The ExecutorService is accepted for signature compatibility and ignored. psf is normalised in place. */
	public static Img< FloatType > convolve( final Img< FloatType > img, final Img< FloatType > psf, final ExecutorService service )
	{
		final long[] d = Buffers.dims( img ), k = Buffers.dims( psf );
		try ( Buffers.Block p = Buffers.toBlock( psf ); Buffers.Slabs src = Buffers.toSlabs( img ); Buffers.Slabs out = new Buffers.Slabs( d[ 0 ], d[ 1 ], d[ 2 ] ) )
		{
			MvsimNative.convolveSlabs( GpuContextPool.get(), src.buffers(), src.nz, d, p.floats, k, 0, out.buffers(), out.nz );
			Buffers.copyBack( p, psf );   // Tools.normImage( psf ) side effect of the reference (:255)
			return Buffers.toImg( out );
		}
	}

	public static Img< FloatType > extractSlices( final RandomAccessibleInterval< FloatType > in, final int inc, final float poissonSNR )
	{
		return extractSlices( in, inc, poissonSNR, rnd );
	}

	public static Img< FloatType > extractSlices( final RandomAccessibleInterval< FloatType > in, final int inc, final float poissonSNR, final Random rnd )
	{
		final long[] d = Buffers.dims( in );
		final long[] o = new long[] { d[ 0 ], d[ 1 ], ( d[ 2 ] - 1 ) / inc + 1 };
		final long seed = poissonSNR >= 0.0 ? rnd.nextLong() : 0L;
		try ( Buffers.Slabs src = Buffers.toSlabs( Views.zeroMin( in ) ); Buffers.Slabs out = new Buffers.Slabs( o[ 0 ], o[ 1 ], o[ 2 ] ) )
		{
			MvsimNative.extractSlicesSlabs( GpuContextPool.get(), src.buffers(), src.nz, d, inc, poissonSNR, seed, 0, out.buffers(), out.nz );
			return Buffers.toImg( out );
		}
	}

	public static Img< FloatType > poissonProcess( final RandomAccessibleInterval< FloatType > in, final float poissonSNR, final Random rnd )
	{
		final long[] d = Buffers.dims( in );
		try ( Buffers.Block b = Buffers.toBlock( Views.zeroMin( in ) ) )
		{
			MvsimNative.poissonProcess( GpuContextPool.get(), b.floats, Buffers.size( d ), poissonSNR, rnd.nextLong(), 0, 0L );
			return Buffers.toImg( b, in.numDimensions() == 3 ? d : new long[] { d[ 0 ], d[ 1 ], 1 } );
		}
	}

	public static Img< FloatType > makeIsotropic( final RandomAccessibleInterval< FloatType > in, final int inc )
	{
		final long[] d = Buffers.dims( in );
		final long[] o = new long[] { d[ 0 ], d[ 1 ], ( d[ 2 ] - 1 ) * inc + 1 };
		try ( Buffers.Block src = Buffers.toBlock( Views.zeroMin( in ) ); Buffers.Block out = Buffers.direct( Buffers.size( o ) ) )
		{
			MvsimNative.makeIsotropic( GpuContextPool.get(), src.floats, d, inc, out.floats );
			return Buffers.toImg( out, o );
		}
	}

	public static Img< FloatType > computeWeightImage( final RandomAccessibleInterval< FloatType > in, final double delta )
	{
		final long[] d = Buffers.dims( in );
		try ( Buffers.Block out = Buffers.direct( Buffers.size( d ) ) )
		{
			MvsimNative.computeWeightImage( GpuContextPool.get(), d, out.floats );
			return Buffers.toImg( out, d );
		}
	}

	/** simulate() of the reference (:366): the shared static generator */
	public static Img< FloatType > simulate() { return simulate( false, rnd ); }

	/** simulate (:366-392): sphere cloud at 2x resolution, then 2x down-sampling; any java.util.Random. */
	public static Img< FloatType > simulate( final boolean halfPixelOffset, final Random rnd )
	{
		final int scale = 2;
		final long n = ( 289 + 1 ) * scale;
		final Img< FloatType > img = new ArrayImgFactory< FloatType >().create( new long[] { n, n, n }, new FloatType() );
		drawSpheres( img, 0, 1, scale, halfPixelOffset, rnd );
		return downSample2x( img );
	}

	/**
	 * drawSpheres (:436-522), in place.  The walk over the large sphere consumes ONE sequential random stream (three draws
	 * per voxel, five where a small sphere starts), so it runs on the host; the ~10^8 voxel updates of the small spheres
	 * are max-composited on the GPU (Math.max makes the order irrelevant).
	 *
	 * ANY java.util.Random works -- SimulateTileStitching passes a plain `new Random( seed )` (:93,:108): the walk then
	 * runs right here with the caller's generator, in the HyperSphereCursor's order (last dimension outermost, nested
	 * truncated radii), and the (centre, radius, value) list goes to mvsim_splat_spheres.  A {@link GpuRandom} is a fast
	 * path only: its 48-bit state crosses the boundary and the native library replays the same walk.
	 */
	public static void drawSpheres( final Img< FloatType > img, final double minValue, final double maxValue, final int scale,
			final boolean halfPixelOffset, final Random rnd )
	{
		final long[] d = Buffers.dims( img );
		try ( Buffers.Block b = Buffers.toBlock( img ) )
		{
			if ( rnd instanceof GpuRandom )
			{
				final GpuRandom g = ( GpuRandom ) rnd;
				final long[] state = new long[] { g.getState() };
				MvsimNative.drawSpheres( GpuContextPool.get(), b.floats, d, minValue, maxValue, scale, halfPixelOffset, state );
				g.setState( state[ 0 ] );
			}
			else
			{
				final long[] c = new long[] { d[ 0 ] / 2, d[ 1 ] / 2, d[ 2 ] / 2 };
				final long R = Math.min( d[ 0 ], Math.min( d[ 1 ], d[ 2 ] ) ) / 2 - 47L * scale - 1;
				if ( R < 0 )
					throw new IllegalArgumentException( "drawSpheres: image too small for scale " + scale );
				final int maxRadius = 10 * scale;
				final long modulus = ( long ) Math.pow( 7 * scale, 3 );            // Util.pow( 7*scale, numDimensions ) (:462)
				final int off = halfPixelOffset ? 1 : 0;
				final List< int[] > geo = new ArrayList<>();
				final List< Float > val = new ArrayList<>();
				for ( long dz = -R; dz <= R; ++dz )
				{
					final long r1 = ( long ) Math.sqrt( R * R - dz * dz );
					for ( long dy = -r1; dy <= r1; ++dy )
					{
						final long r0 = ( long ) Math.sqrt( r1 * r1 - dy * dy );
						for ( long dx = -r0; dx <= r0; ++dx )
						{
							final int radius = rnd.nextInt( maxRadius ) + 1;               // :468
							final double randomValue = rnd.nextDouble();                    // :475
							if ( Math.round( randomValue * 10000 ) % modulus != 0 )         // :478
								continue;
							final double value = rnd.nextDouble() * ( maxValue - minValue ) + minValue;   // :480
							geo.add( new int[] { ( int ) ( c[ 0 ] + dx + off ), ( int ) ( c[ 1 ] + dy + off ), ( int ) ( c[ 2 ] + dz ), radius } );
							val.add( ( float ) value );
						}
					}
				}
				final int[] g4 = new int[ 4 * geo.size() ];
				final float[] v = new float[ geo.size() ];
				for ( int i = 0; i < v.length; ++i )
				{
					System.arraycopy( geo.get( i ), 0, g4, 4 * i, 4 );
					v[ i ] = val.get( i );
				}
				MvsimNative.splatSpheres( GpuContextPool.get(), b.floats, d, g4, v );
			}
			Buffers.copyBack( b, img );
		}
	}

	/** downSample2x (:394-424) */
	public static Img< FloatType > downSample2x( final RandomAccessibleInterval< FloatType > in )
	{
		final long[] d = Buffers.dims( in );
		final long[] o = new long[] { d[ 0 ] / 2 - 1, d[ 1 ] / 2 - 1, d[ 2 ] / 2 - 1 };
		try ( Buffers.Block src = Buffers.toBlock( Views.zeroMin( in ) ); Buffers.Block out = Buffers.direct( Buffers.size( o ) ) )
		{
			MvsimNative.downSample2x( GpuContextPool.get(), src.floats, d, out.floats );
			return Buffers.toImg( out, o );
		}
	}

	/**
	 * The block of main() after the view loop (:615-640): per voxel, sum the weights of all views; zero sum gives
	 * zero weights, otherwise w = min( 1, osem * w / sum ).  In place on the given images.
	 */
	public static void normalizeWeights( final List< Img< FloatType > > weights, final float osem )
	{
		final Buffers.Block[] b = new Buffers.Block[ weights.size() ];
		try
		{
			final java.nio.FloatBuffer[] f = new java.nio.FloatBuffer[ b.length ];
			for ( int i = 0; i < b.length; ++i )
			{
				b[ i ] = Buffers.toBlock( weights.get( i ) );
				f[ i ] = b[ i ].floats;
			}
			MvsimNative.normalizeWeights( GpuContextPool.get(), f, Buffers.size( Buffers.dims( weights.get( 0 ) ) ), osem );
			for ( int i = 0; i < b.length; ++i )
				Buffers.copyBack( b[ i ], weights.get( i ) );
		}
		finally
		{
			for ( final Buffers.Block x : b )
				if ( x != null ) x.close();
		}
	}

	/** Which stages of a view come back to the host (the acquisition always does). */
	public static final int ROT = 1, ATT = 2, CON = 4;

	/**
	 * One view of the main loop (:570-585) as ONE native call, intermediates resident in HBM.  `want` is a bit set of
	 * {@link #ROT}, {@link #ATT}, {@link #CON}: only those volumes are materialised and copied back (each costs a 0.5 GB
	 * PCIe transfer at 512^3, against 2.4 ms for the whole view on the GPU).  Returns { rot, att, con, acq } with null
	 * for stages that were not asked for.
	 */
	public static Img< FloatType >[] simulateView( final RandomAccessibleInterval< FloatType > groundTruth, final Img< FloatType > psf,
			final int degrees, final double attenuation, final int lightsheetSpacing, final float poissonSNR, final Random rnd, final int view,
			final int want )
	{
		final long[] d = Buffers.dims( groundTruth ), k = Buffers.dims( psf );
		final long[] o = new long[] { d[ 0 ], d[ 1 ], ( d[ 2 ] - 1 ) / lightsheetSpacing + 1 };
		final long n = Buffers.size( d );
		try ( Buffers.Block p = Buffers.toBlock( psf ); Buffers.Block gt = Buffers.toBlock( Views.zeroMin( groundTruth ) );
				Buffers.Block rot = ( want & ROT ) != 0 ? Buffers.direct( n ) : null; Buffers.Block att = ( want & ATT ) != 0 ? Buffers.direct( n ) : null;
				Buffers.Block con = ( want & CON ) != 0 ? Buffers.direct( n ) : null; Buffers.Block acq = Buffers.direct( Buffers.size( o ) ) )
		{
			MvsimNative.simulateView( GpuContextPool.get(), gt.floats, d, p.floats, k, 0, degrees, attenuation, minValue, avgIntensity,
					lightsheetSpacing, poissonSNR, rnd.nextLong(), view, rot != null ? rot.floats : null, att != null ? att.floats : null,
					con != null ? con.floats : null, acq.floats );
			Buffers.copyBack( p, psf );
			@SuppressWarnings( "unchecked" )
			final Img< FloatType >[] res = new Img[] { rot != null ? Buffers.toImg( rot, d ) : null, att != null ? Buffers.toImg( att, d ) : null,
					con != null ? Buffers.toImg( con, d ) : null, Buffers.toImg( acq, o ) };
			return res;
		}
	}

	/** as above with every intermediate (what main() saves, :598-604) */
	public static Img< FloatType >[] simulateView( final RandomAccessibleInterval< FloatType > groundTruth, final Img< FloatType > psf,
			final int degrees, final double attenuation, final int lightsheetSpacing, final float poissonSNR, final Random rnd, final int view )
	{
		return simulateView( groundTruth, psf, degrees, attenuation, lightsheetSpacing, poissonSNR, rnd, view, ROT | ATT | CON );
	}

	/**
	 * The view loop of main() (:567-585) pipelined over PCIe: every view is submitted with mvsim_simulate_view_async, which
	 * returns at once, and the previous view is collected while the next one uploads and computes (two page-locked staging
	 * sets inside the library, three HIP streams).  The ground truth crosses PCIe ONCE (same buffer for every angle); per
	 * view only the acquisition comes back.  psfs.get( v ) is normalised in place like convolve does.
	 */
	private static boolean sameDimensions( final List< Img< FloatType > > imgs )
	{
		for ( final Img< FloatType > i : imgs )
			for ( int k = 0; k < 3; ++k )
				if ( Buffers.dims( i )[ k ] != Buffers.dims( imgs.get( 0 ) )[ k ] )
					return false;
		return true;
	}

	private static List< Img< FloatType > > simulateViewsBatched( final long ctx, final RandomAccessibleInterval< FloatType > groundTruth,
			final List< Img< FloatType > > psfs, final int[] degrees, final double attenuation, final int lightsheetSpacing, final float poissonSNR,
			final Random rnd, final long[] d, final long[] o )
	{
		final int n = degrees.length;
		final Buffers.Block[] acq = new Buffers.Block[ n ], psf = new Buffers.Block[ n ];
		final java.nio.FloatBuffer[] pb = new java.nio.FloatBuffer[ n ], ab = new java.nio.FloatBuffer[ n ];
		final long[] seeds = new long[ n ];
		try ( Buffers.Block gt = Buffers.toBlock( Views.zeroMin( groundTruth ) ) )
		{
			for ( int v = 0; v < n; ++v )
			{
				psf[ v ] = Buffers.toBlock( psfs.get( v ) );
				acq[ v ] = Buffers.direct( Buffers.size( o ) );
				pb[ v ] = psf[ v ].floats;
				ab[ v ] = acq[ v ].floats;
				seeds[ v ] = rnd.nextLong();                              // one draw per view, in view order, as the pipelined form does
			}
			MvsimNative.simulateViewsBatch( ctx, gt.floats, d, pb, Buffers.dims( psfs.get( 0 ) ), degrees, attenuation, minValue, avgIntensity,
					lightsheetSpacing, poissonSNR, seeds, ab );
			final List< Img< FloatType > > result = new ArrayList<>();
			for ( int v = 0; v < n; ++v )
			{
				result.add( Buffers.toImg( acq[ v ], o ) );
				Buffers.copyBack( psf[ v ], psfs.get( v ) );              // normalised in place, as convolve() leaves it (Q5)
			}
			return result;
		}
		finally
		{
			for ( int v = 0; v < n; ++v )
			{
				if ( acq[ v ] != null ) acq[ v ].close();
				if ( psf[ v ] != null ) psf[ v ].close();
			}
		}
	}

	public static List< Img< FloatType > > simulateViews( final RandomAccessibleInterval< FloatType > groundTruth, final List< Img< FloatType > > psfs,
			final int[] degrees, final double attenuation, final int lightsheetSpacing, final float poissonSNR, final Random rnd )
	{
		final long[] d = Buffers.dims( groundTruth );
		final long[] o = new long[] { d[ 0 ], d[ 1 ], ( d[ 2 ] - 1 ) / lightsheetSpacing + 1 };
		final long ctx = GpuContextPool.get();
		final List< Img< FloatType > > result = new ArrayList<>();
		// Views that cannot fill the chip one at a time -- the reference's own run: seven views of a 289^3 volume (:376-380, :567) -- go to
		// the library in ONE call: it stacks them (one kernel launch per stage for all views) and brings the counts back as 16-bit values.
		if ( Buffers.size( d ) <= ( 1L << 26 ) && degrees.length > 1 && degrees.length <= 32 && sameDimensions( psfs ) )
			return simulateViewsBatched( ctx, groundTruth, psfs, degrees, attenuation, lightsheetSpacing, poissonSNR, rnd, d, o );
		final Buffers.Block[] acq = new Buffers.Block[ 2 ], psf = new Buffers.Block[ 2 ];
		final long[] ticket = new long[ 2 ];
		try ( Buffers.Block gt = Buffers.toBlock( Views.zeroMin( groundTruth ) ) )
		{
			for ( int v = 0; v < degrees.length; ++v )
			{
				final int s = v & 1;
				if ( acq[ s ] != null )                                   // view v - 2 used this staging pair: collect it first
				{
					MvsimNative.waitView( ctx, ticket[ s ] );
					result.add( Buffers.toImg( acq[ s ], o ) );
					Buffers.copyBack( psf[ s ], psfs.get( v - 2 ) );
					acq[ s ].close(); psf[ s ].close();
					acq[ s ] = null; psf[ s ] = null;
				}
				psf[ s ] = Buffers.toBlock( psfs.get( v ) );
				acq[ s ] = Buffers.direct( Buffers.size( o ) );
				ticket[ s ] = MvsimNative.simulateViewAsync( ctx, gt.floats, gt.generation, d, psf[ s ].floats, Buffers.dims( psfs.get( v ) ), 0, degrees[ v ],
						attenuation, minValue, avgIntensity, lightsheetSpacing, poissonSNR, rnd.nextLong(), v, acq[ s ].floats );
			}
			// the last two views, in order
			for ( int v = Math.max( 0, degrees.length - 2 ); v < degrees.length; ++v )
			{
				final int s = v & 1;
				MvsimNative.waitView( ctx, ticket[ s ] );
				result.add( Buffers.toImg( acq[ s ], o ) );
				Buffers.copyBack( psf[ s ], psfs.get( v ) );
				acq[ s ].close(); psf[ s ].close();
				acq[ s ] = null; psf[ s ] = null;
			}
			return result;
		}
		finally
		{
			for ( int s = 0; s < 2; ++s )
			{
				if ( acq[ s ] != null ) { MvsimNative.waitViewQuiet( ctx, ticket[ s ] ); acq[ s ].close(); }
				if ( psf[ s ] != null ) psf[ s ].close();
			}
		}
	}

	/**
	 * The driver of the reference, SimulateMultiViewDataset.main (:524-665), on the GPU operators: same hard-coded parameters
	 * (:531-548: SNR 25, light-sheet spacing 3, attenuation 0.01f, seven angles at 52 degree steps, OSEM 3, angle offset 15),
	 * same stages per angle, same output files next to the PSF stacks (:526, :561-562, :598-604, :645, :663).  File I/O and the
	 * window toolkit stay the reference's own (net.preibisch.simulation.Tools.open / save, ij.ImageJ).
	 */
	public static void main( final String[] args )
	{
		final String dir = "src/main/resources/";
		final java.util.concurrent.ExecutorService service = java.util.concurrent.Executors.newFixedThreadPool( Runtime.getRuntime().availableProcessors() );

		new ij.ImageJ();

		final float poissonSNR = 25f;
		final int lightsheetSpacing = 3;
		final float attenuation = 0.01f;
		final int angleIncrement = 52;       // seven angles
		final float osem = 3.0f;
		final int angleOffset = 15;          // so that everything is rotated at least once

		System.out.println( new java.util.Date( System.currentTimeMillis() ) + ": rendering basis for ground truth" );
		final Img< FloatType > rendered = simulate( false, rnd );

		System.out.println( new java.util.Date( System.currentTimeMillis() ) + ": computing ground truth" );
		final Img< FloatType > obj = rotateAroundAxis( rendered, 0, angleOffset );

		net.preibisch.simulation.Tools.save( rendered, dir + "rendered.tif" );
		net.preibisch.simulation.Tools.save( obj, dir + "groundtruth.tif" );

		final ArrayList< Img< FloatType > > weights = new ArrayList< Img< FloatType > >();

		for ( int angle = 0; angle < 360; angle += angleIncrement )
		{
			System.out.println( new java.util.Date( System.currentTimeMillis() ) + ": view at angle " + angle );
			final Img< FloatType > psf = net.preibisch.simulation.Tools.open( dir + "Angle" + angle + ".tif", true );
			// rotate, attenuate, convolve, adjustImage, extractSlices (:570-585) as ONE native call, every intermediate kept
			final Img< FloatType >[] v = simulateView( rendered, psf, angle + angleOffset, attenuation, lightsheetSpacing, poissonSNR, rnd, angle / angleIncrement );
			final Img< FloatType > rot = v[ 0 ], att = v[ 1 ], con = v[ 2 ], acq = v[ 3 ];
			final Img< FloatType > w = computeWeightImage( rot, attenuation );
			final Img< FloatType > iso = makeIsotropic( acq, lightsheetSpacing );
			final Img< FloatType > view = rotateAroundAxis( iso, 0, -angle );
			final Img< FloatType > viewWeights = rotateAroundAxis( w, 0, -angle );
			final Img< FloatType > viewPSF = rotateAroundAxis( psf, 0, -angle );

			net.preibisch.simulation.Tools.save( rot, dir + "rot_view_" + angle + ".tif" );
			net.preibisch.simulation.Tools.save( att, dir + "att_view_" + angle + ".tif" );
			net.preibisch.simulation.Tools.save( con, dir + "con_view_" + angle + ".tif" );
			net.preibisch.simulation.Tools.save( acq, dir + "acq_view_" + angle + ".tif" );
			net.preibisch.simulation.Tools.save( iso, dir + "iso_view_" + angle + ".tif" );
			net.preibisch.simulation.Tools.save( view, dir + "aligned_view_" + angle + ".tif" );
			net.preibisch.simulation.Tools.save( viewPSF, dir + "aligned_view_psf_" + angle + ".tif" );

			weights.add( viewWeights );
		}

		// norm sum weights to osem (:615-640)
		normalizeWeights( weights, osem );

		for ( int i = 0; i < weights.size(); ++i )
			net.preibisch.simulation.Tools.save( weights.get( i ), dir + "aligned_view_weights" + ( i * angleIncrement ) + ".tif" );

		// sum of the normalised weights (:647-663)
		final Img< FloatType > sumWeights = weights.get( 0 ).factory().create( weights.get( 0 ), weights.get( 0 ).firstElement() );
		for ( final Img< FloatType > w : weights )
		{
			final net.imglib2.Cursor< FloatType > s = sumWeights.cursor();
			final net.imglib2.Cursor< FloatType > c = w.cursor();
			while ( s.hasNext() )
				s.next().add( c.next() );
		}
		net.preibisch.simulation.Tools.save( sumWeights, dir + "sum_weights.tif" );
		service.shutdown();
		System.out.println( "done" );
	}
}
