package net.preibisch.simulation.gpu;

import java.nio.ByteBuffer;
import java.nio.ByteOrder;
import java.nio.FloatBuffer;

import net.imglib2.Cursor;
import net.imglib2.Interval;
import net.imglib2.RandomAccessibleInterval;
import net.imglib2.img.Img;
import net.imglib2.img.array.ArrayImg;
import net.imglib2.img.array.ArrayImgs;
import net.imglib2.img.basictypeaccess.array.FloatArray;
import net.imglib2.type.numeric.real.FloatType;
import net.imglib2.view.Views;

/** Marshalling between ImgLib2 images and the flat x-fastest float32 layout of the C ABI. */
final class Buffers
{
	private Buffers() {}

	/**
	 * Staging buffers come from a small per-thread pool of PAGE-LOCKED blocks (mvsim_host_alloc wrapped by
	 * NewDirectByteBuffer): copies between such a block and HBM run at PCIe speed (30 ms per 512^3 view against 50 ms
	 * from pageable memory), and page-locking is slow, so blocks are kept and reused by size instead of being
	 * allocated per call.  Falls back to an ordinary direct buffer when the native allocation fails.
	 */
	private static final ThreadLocal< java.util.HashMap< Long, java.util.ArrayDeque< ByteBuffer > > > POOL =
			ThreadLocal.withInitial( java.util.HashMap::new );

	static FloatBuffer direct( final long n )
	{
		if ( n > Integer.MAX_VALUE / 4 )
			throw new IllegalArgumentException( "image has more than 2^29 voxels: pass z-slabs (see INTEGRATION.md)" );
		final long bytes = 4 * n;
		final java.util.ArrayDeque< ByteBuffer > free = POOL.get().get( bytes );
		ByteBuffer b = free != null ? free.poll() : null;
		if ( b == null )
			b = MvsimNative.allocPinned( GpuContextPool.get(), bytes );
		if ( b == null )
			b = ByteBuffer.allocateDirect( (int)bytes );
		b.clear();
		return b.order( ByteOrder.nativeOrder() ).asFloatBuffer();
	}

	/** hand a staging block back to the pool once its contents have been copied into an Img */
	static void recycle( final ByteBuffer block )
	{
		POOL.get().computeIfAbsent( ( long )block.capacity(), k -> new java.util.ArrayDeque<>() ).add( block );
	}

	static long[] dims( final Interval i )
	{
		final long[] d = new long[ 3 ];
		for ( int k = 0; k < 3; ++k )
			d[ k ] = k < i.numDimensions() ? i.dimension( k ) : 1;
		return d;
	}

	static long size( final long[] d ) { return d[ 0 ] * d[ 1 ] * d[ 2 ]; }

	/** Any RAI view (zero-min or not, SimulateTileStitching.java:152-156) -> direct buffer in flat iteration order. */
	static FloatBuffer toBuffer( final RandomAccessibleInterval< FloatType > rai )
	{
		final FloatBuffer b = direct( size( dims( rai ) ) );
		if ( rai instanceof ArrayImg && ( (ArrayImg< ?, ? >)rai ).update( null ) instanceof FloatArray )
		{
			b.put( ( (FloatArray)( (ArrayImg< ?, ? >)rai ).update( null ) ).getCurrentStorageArray() );
		}
		else
		{
			final Cursor< FloatType > c = Views.flatIterable( rai ).cursor();
			while ( c.hasNext() )
				b.put( c.next().get() );
		}
		b.rewind();
		return b;
	}

	static Img< FloatType > toImg( final FloatBuffer b, final long[] d )
	{
		final float[] a = new float[ (int)size( d ) ];
		b.rewind();
		b.get( a );
		return ArrayImgs.floats( a, d );
	}

	/** Copy results back into a caller-owned image (in-place operators: normImage, adjustImage, poissonProcess). */
	static void copyBack( final FloatBuffer b, final Iterable< FloatType > img )
	{
		b.rewind();
		for ( final FloatType t : img )
			t.set( b.get() );
	}
}
