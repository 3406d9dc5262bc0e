package net.preibisch.simulation.gpu;

import java.nio.ByteBuffer;
import java.nio.ByteOrder;
import java.nio.FloatBuffer;
import java.util.ArrayDeque;
import java.util.HashMap;

import net.imglib2.Cursor;
import net.imglib2.Interval;
import net.imglib2.RandomAccessibleInterval;
import net.imglib2.img.Img;
import net.imglib2.img.array.ArrayImg;
import net.imglib2.img.array.ArrayImgs;
import net.imglib2.img.basictypeaccess.array.FloatArray;
import net.imglib2.type.numeric.real.FloatType;
import net.imglib2.view.Views;

/**
 * Marshalling between ImgLib2 images and the flat x-fastest float32 layout of the C ABI.
 *
 * Staging memory comes as {@link Block}s: PAGE-LOCKED bytes (mvsim_host_alloc wrapped by NewDirectByteBuffer; copies
 * between such a block and HBM run at PCIe speed) together with their float view.  A block is OWNED by whoever asked
 * for it and must be closed (try-with-resources in every facade method): close() hands it back to a small per-thread
 * pool keyed by size -- page-locking is slow, so blocks are reused -- and the pool is capped: beyond
 * {@code -Dmvsim.pool.bytes} (default 4 GiB) per thread a closed block is freed at once (mvsim_host_free).  The JVM never
 * frees NewDirectByteBuffer memory it did not allocate, so nothing here relies on the garbage collector; the pool itself
 * is emptied when the thread's native context goes away ({@link GpuContextPool}).
 */
final class Buffers
{
	private Buffers() {}

	static final class Block implements AutoCloseable
	{
		final ByteBuffer bytes;
		final FloatBuffer floats;
		final boolean pinned;
		private boolean open = true;

		Block( final ByteBuffer bytes, final boolean pinned )
		{
			this.bytes = bytes;
			this.pinned = pinned;
			bytes.clear();
			this.floats = bytes.order( ByteOrder.nativeOrder() ).asFloatBuffer();
		}

		@Override
		public void close()
		{
			if ( open )
			{
				open = false;
				recycle( this );
			}
		}
	}

	private static final class Pool
	{
		final HashMap< Long, ArrayDeque< ByteBuffer > > free = new HashMap<>();
		long pooledBytes = 0;
	}

	private static final ThreadLocal< Pool > POOL = ThreadLocal.withInitial( Pool::new );
	private static final long POOL_CAP = Long.getLong( "mvsim.pool.bytes", 4L << 30 );

	/** a staging block of n floats (page-locked when the native allocation succeeds) */
	static Block direct( final long n )
	{
		if ( n > Integer.MAX_VALUE / 4 )
			throw new IllegalArgumentException( "image has more than 2^29 voxels: pass z-slabs (simulateViewSlabs, INTEGRATION.md)" );
		final long bytes = 4 * Math.max( n, 1 );
		final Pool pool = POOL.get();
		final ArrayDeque< ByteBuffer > q = pool.free.get( bytes );
		ByteBuffer b = q != null ? q.poll() : null;
		if ( b != null )
		{
			pool.pooledBytes -= bytes;
			return new Block( b, true );
		}
		b = MvsimNative.allocPinned( GpuContextPool.get(), bytes );
		if ( b != null )
			return new Block( b, true );
		return new Block( ByteBuffer.allocateDirect( ( int ) bytes ), false );    // ordinary direct memory: the GC owns it
	}

	private static void recycle( final Block block )
	{
		if ( !block.pinned )
			return;
		final Pool pool = POOL.get();
		final long bytes = block.bytes.capacity();
		if ( pool.pooledBytes + bytes > POOL_CAP )
		{
			MvsimNative.freePinned( block.bytes );
			return;
		}
		pool.free.computeIfAbsent( bytes, k -> new ArrayDeque<>() ).add( block.bytes );
		pool.pooledBytes += bytes;
	}

	/** free every pooled block of the calling thread (called when its native context is destroyed) */
	static void drain()
	{
		final Pool pool = POOL.get();
		for ( final ArrayDeque< ByteBuffer > q : pool.free.values() )
			for ( final ByteBuffer b : q )
				MvsimNative.freePinned( b );
		pool.free.clear();
		pool.pooledBytes = 0;
	}

	static long[] dims( final Interval i )
	{
		final long[] d = new long[ 3 ];
		for ( int k = 0; k < 3; ++k )
			d[ k ] = k < i.numDimensions() ? i.dimension( k ) : 1;
		return d;
	}

	static long size( final long[] d ) { return d[ 0 ] * d[ 1 ] * d[ 2 ]; }

	/** Any RAI view (zero-min or not, SimulateTileStitching.java:152-156) -> staging block in flat iteration order. */
	static Block toBlock( final RandomAccessibleInterval< FloatType > rai )
	{
		final Block blk = direct( size( dims( rai ) ) );
		final FloatBuffer b = blk.floats;
		try
		{
			if ( rai instanceof ArrayImg && ( ( ArrayImg< ?, ? > ) rai ).update( null ) instanceof FloatArray )
			{
				b.put( ( ( FloatArray ) ( ( ArrayImg< ?, ? > ) rai ).update( null ) ).getCurrentStorageArray() );
			}
			else
			{
				final Cursor< FloatType > c = Views.flatIterable( rai ).cursor();
				while ( c.hasNext() )
					b.put( c.next().get() );
			}
			b.rewind();
			return blk;
		}
		catch ( final RuntimeException e )
		{
			blk.close();
			throw e;
		}
	}

	/** a fresh ArrayImg holding the block's first size(d) floats (the block stays owned by the caller) */
	static Img< FloatType > toImg( final Block blk, final long[] d )
	{
		final float[] a = new float[ ( int ) size( d ) ];
		blk.floats.rewind();
		blk.floats.get( a );
		return ArrayImgs.floats( a, d );
	}

	/** Copy results back into a caller-owned image (in-place operators: normImage, adjustImage, poissonProcess). */
	static void copyBack( final Block blk, final Iterable< FloatType > img )
	{
		blk.floats.rewind();
		for ( final FloatType t : img )
			t.set( blk.floats.get() );
	}
}
