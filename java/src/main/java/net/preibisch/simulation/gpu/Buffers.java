package net.preibisch.simulation.gpu;

import java.nio.ByteBuffer;
import java.nio.ByteOrder;
import java.nio.FloatBuffer;
import java.util.ArrayDeque;
import java.util.ArrayList;
import java.util.HashMap;
import java.util.List;
import java.util.concurrent.atomic.AtomicLong;

import net.imglib2.Cursor;
import net.imglib2.Interval;
import net.imglib2.RandomAccessibleInterval;
import net.imglib2.img.Img;
import net.imglib2.img.array.ArrayImg;
import net.imglib2.img.array.ArrayImgs;
import net.imglib2.img.basictypeaccess.array.FloatArray;
import net.imglib2.img.basictypeaccess.nio.FloatBufferAccess;
import net.imglib2.type.numeric.real.FloatType;
import net.imglib2.util.Fraction;
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
 *
 * What an operator RETURNS (round 6).  The reference's operators return a new ArrayImg (SimulateMultiViewDataset.java:109,198,235,256,
 * 321), and so does the facade -- by default a heap-backed one, filled from the staging block by {@code MvsimNative.copyFloats}: the
 * array is held with GetPrimitiveArrayCritical and the library's host threads copy (and take the fresh array's first-touch page
 * faults) in parallel; one JVM thread needs 79 ms for a 512^3 image that way, 44 x the GPU time of the view it returns
 * (profiles/r05_slab_copy.txt, r06_slab_copy.txt).  With {@code -Dmvsim.zero_copy=true} the returned image is an
 * {@code ArrayImg<FloatType, FloatBufferAccess>} OVER the page-locked block itself: no copy at all; the block then belongs to the
 * image and is freed (mvsim_host_free) by a {@link java.lang.ref.Cleaner} once the image is unreachable.  Every ImgLib2 accessor
 * works on it; only code that unwraps {@code FloatArray} storage by hand does not -- hence opt-in.  Images of more than one block
 * (> 2^28 floats) are always copied.
 */
final class Buffers
{
	private Buffers() {}

	/**
	 * Every time a block is (re)filled with image data it gets a new generation.  mvsim_simulate_view_async skips the upload of
	 * a ground truth whose host pointer AND generation it has seen already; blocks are pooled by size, so a second dataset of
	 * the same size lands at the SAME host address -- only the generation tells the library that the contents changed.
	 */
	private static final AtomicLong GENERATION = new AtomicLong( 1 );

	/** floats per z-slab block of a large volume: one direct ByteBuffer holds at most 2^31 - 1 bytes */
	static final long MAX_BLOCK_FLOATS = 1L << 28;

	/** return images over the page-locked staging block instead of copying it into a heap array (see the class comment) */
	static final boolean ZERO_COPY = Boolean.getBoolean( "mvsim.zero_copy" );
	/** from this many floats up the bulk copies between heap arrays and blocks run on the native host threads */
	static final long NATIVE_COPY_MIN = 1L << 20;
	private static final java.lang.ref.Cleaner CLEANER = java.lang.ref.Cleaner.create();

	static final class Block implements AutoCloseable
	{
		final ByteBuffer bytes;
		final FloatBuffer floats;
		final boolean pinned;
		/** identity of the contents (see {@link #GENERATION}); 0 for a block that holds no image data yet */
		long generation = 0;
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

		/** the block now belongs to an image that wraps it ({@link #wrap}): close() must not hand it back to the pool */
		void detach()
		{
			open = false;
		}
	}

	/** heap array <-> block, on the native host threads from {@link #NATIVE_COPY_MIN} floats up (else the JVM's own bulk copy) */
	private static void copy( final Block blk, final long blockOff, final float[] a, final int arrayOff, final int n, final boolean toArray )
	{
		if ( n >= NATIVE_COPY_MIN && blk.pinned )
		{
			MvsimNative.copyFloats( GpuContextPool.get(), blk.floats, blockOff, a, arrayOff, n, toArray );
			return;
		}
		final FloatBuffer b = blk.floats.duplicate();
		b.position( ( int ) blockOff );
		if ( toArray )
			b.get( a, arrayOff, n );
		else
			b.put( a, arrayOff, n );
	}

	/**
	 * An image OVER the block's first size(d) floats, no copy: the block belongs to the image from here on and is freed when the
	 * image is unreachable.  The cleaning action holds the bytes, never the image.
	 */
	private static Img< FloatType > wrap( final Block blk, final long[] d )
	{
		final FloatBuffer fb = blk.floats.duplicate();
		fb.rewind();
		fb.limit( ( int ) size( d ) );
		final ArrayImg< FloatType, FloatBufferAccess > img = new ArrayImg<>( new FloatBufferAccess( fb.slice(), true ), d, new Fraction() );
		img.setLinkedType( new FloatType( img ) );
		final ByteBuffer bytes = blk.bytes;
		final boolean pinned = blk.pinned;
		blk.detach();
		if ( pinned )
			CLEANER.register( img, () -> MvsimNative.freePinned( bytes ) );
		return img;
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
			throw new IllegalArgumentException( "one staging block holds at most 2^29 - 1 floats: larger volumes travel as z slabs (Buffers.Slabs)" );
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
				final float[] a = ( ( FloatArray ) ( ( ArrayImg< ?, ? > ) rai ).update( null ) ).getCurrentStorageArray();
				copy( blk, 0, a, 0, a.length, false );
			}
			else
			{
				final Cursor< FloatType > c = Views.flatIterable( rai ).cursor();
				while ( c.hasNext() )
					b.put( c.next().get() );
			}
			b.rewind();
			blk.generation = GENERATION.getAndIncrement();
			return blk;
		}
		catch ( final RuntimeException e )
		{
			blk.close();
			throw e;
		}
	}

	/**
	 * A volume as a list of z-slab blocks (whole planes each, at most {@link #MAX_BLOCK_FLOATS} floats per block): the form in
	 * which the per-stage operators hand images to the native side.  A 512^3 image is ONE slab; a 1024^3 ArrayImg (2^30
	 * voxels: fine for ImgLib2, SimulateMultiViewDataset.java:109, but four times what one direct buffer holds) becomes four.
	 */
	static final class Slabs implements AutoCloseable
	{
		final List< Block > blocks = new ArrayList<>();
		final long[] nz;            // planes per slab
		final long[] dim;           // { nx, ny, planes in total }

		Slabs( final long nx, final long ny, final long planes )
		{
			dim = new long[] { nx, ny, planes };
			final long plane = nx * ny;
			if ( plane > MAX_BLOCK_FLOATS )
				throw new IllegalArgumentException( "a single plane of " + plane + " voxels exceeds one staging block" );
			final long per = Math.max( 1, MAX_BLOCK_FLOATS / plane );
			final int count = ( int ) ( ( planes + per - 1 ) / per );
			nz = new long[ count ];
			try
			{
				for ( int i = 0; i < count; ++i )
				{
					nz[ i ] = Math.min( per, planes - i * per );
					blocks.add( direct( plane * nz[ i ] ) );
				}
			}
			catch ( final RuntimeException e )
			{
				close();
				throw e;
			}
		}

		FloatBuffer[] buffers()
		{
			final FloatBuffer[] f = new FloatBuffer[ blocks.size() ];
			for ( int i = 0; i < f.length; ++i )
				f[ i ] = blocks.get( i ).floats;
			return f;
		}

		@Override
		public void close()
		{
			for ( final Block b : blocks )
				b.close();
			blocks.clear();
		}
	}

	/** Any RAI view -> z-slab blocks in flat iteration order (x fastest). */
	static Slabs toSlabs( final RandomAccessibleInterval< FloatType > rai )
	{
		final long[] d = dims( rai );
		final Slabs s = new Slabs( d[ 0 ], d[ 1 ], d[ 2 ] );
		try
		{
			final long plane = d[ 0 ] * d[ 1 ];
			if ( rai instanceof ArrayImg && ( ( ArrayImg< ?, ? > ) rai ).update( null ) instanceof FloatArray )
			{
				final float[] a = ( ( FloatArray ) ( ( ArrayImg< ?, ? > ) rai ).update( null ) ).getCurrentStorageArray();
				long off = 0;
				for ( int i = 0; i < s.nz.length; ++i )
				{
					final long n = plane * s.nz[ i ];
					copy( s.blocks.get( i ), 0, a, ( int ) off, ( int ) n, false );       // off + n <= a.length < 2^31
					s.blocks.get( i ).floats.rewind();
					off += n;
				}
			}
			else
			{
				final Cursor< FloatType > c = Views.flatIterable( rai ).cursor();
				for ( int i = 0; i < s.nz.length; ++i )
				{
					final FloatBuffer b = s.blocks.get( i ).floats;
					for ( long k = plane * s.nz[ i ]; k > 0; --k )
						b.put( c.next().get() );
					b.rewind();
				}
			}
			final long g = GENERATION.getAndIncrement();
			for ( final Block b : s.blocks )
				b.generation = g;
			return s;
		}
		catch ( final RuntimeException e )
		{
			s.close();
			throw e;
		}
	}

	/** a fresh ArrayImg holding the slabs' planes (up to 2^31 - 1 voxels: the limit of ArrayImg itself) */
	static Img< FloatType > toImg( final Slabs s )
	{
		final long n = size( s.dim );
		if ( n > Integer.MAX_VALUE )
			throw new IllegalArgumentException( "an ArrayImg holds at most 2^31 - 1 voxels (the reference has the same limit)" );
		if ( ZERO_COPY && s.blocks.size() == 1 )
			return wrap( s.blocks.get( 0 ), s.dim );               // (s.close() then finds the block detached)
		final float[] a = new float[ ( int ) n ];
		final long plane = s.dim[ 0 ] * s.dim[ 1 ];
		long off = 0;
		for ( int i = 0; i < s.nz.length; ++i )
		{
			copy( s.blocks.get( i ), 0, a, ( int ) off, ( int ) ( plane * s.nz[ i ] ), true );
			off += plane * s.nz[ i ];
		}
		return ArrayImgs.floats( a, s.dim );
	}

	/**
	 * An ArrayImg holding the block's first size(d) floats: a fresh heap-backed one (the block stays owned by the caller), or, with
	 * {@link #ZERO_COPY}, one over the block itself (the caller's close() of the block is then a no-op).
	 */
	static Img< FloatType > toImg( final Block blk, final long[] d )
	{
		if ( ZERO_COPY )
			return wrap( blk, d );
		final float[] a = new float[ ( int ) size( d ) ];
		copy( blk, 0, a, 0, a.length, true );
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
