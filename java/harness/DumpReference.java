/*
 * Pins the CPU oracle (oracle/mvsim_oracle.c) and the HIP path to the REFERENCE itself.
 *
 * This program is not part of the product and contains nothing of the reference: it CALLS
 * net.preibisch.simulation.SimulateMultiViewDataset / Tools (reference jar on the classpath) on the raw
 * float32 inputs that tests/golden/make_golden.py writes to tests/golden/ref_in/ and dumps what they return as raw
 * little-endian float32 / float64 files into tests/golden/ref/.  tests/test_reference_dumps.py then compares the
 * oracle (CPU suite) and libmvsim (GPU suite) with those files; without them the tests skip and parity stays
 * "unpinned" (DESIGN.md section 2).  Nobody could run it in the build container (no JDK there).
 *
 *   mvn -f java/pom.xml -Pharness package            # or: make -C java harness REF_CP=<classpath of the reference>
 *   java -cp java/target/classes:<reference classpath> DumpReference tests/golden/ref_in tests/golden/ref
 *
 * manifest.txt (one case per line, written by make_golden.py):
 *   name nx ny nz kx ky kz axis degrees delta inc minValue targetAverage snr seed
 * inputs per case: <name>.gt.raw (nx*ny*nz float32, x fastest), <name>.psf.raw (kx*ky*kz float32, un-normalised)
 * outputs per case (ArrayImg order = x fastest, SimulateMultiViewDataset.java:115-132):
 *   <name>.affine.f64     12 doubles, row major 3x4, read off the model's action on 0, ex, ey, ez (differences: exact to ~1e-13)   axisRotation  :80-102
 *   <name>.rot.raw                                          rotateAroundAxis             :104-135
 *   <name>.att.raw                                          attenuate3d                  :318-364
 *   <name>.psf_norm.raw   the PSF after convolve() normalised it in place (Q5)           :255, Tools.java:112-118
 *   <name>.con_raw.raw                                      convolve                     :253-264
 *   <name>.con.raw        after Tools.adjustImage           Tools.java:143-159
 *   <name>.corr.f64       1 double: adjustImage's return value
 *   <name>.ext.raw        extractSlices(con, inc, -1)       :181-231 (no noise: bit-exact copy)
 *   <name>.iso.raw        makeIsotropic(ext, inc)           :144-171
 *   <name>.weight.raw     computeWeightImage(iso, delta)    :280-316
 *   <name>.poisson.raw    Tools.poissonProcess(copy of ext, snr, new Random(seed))   Tools.java:73-86 (the reference's own sampler)
 */
import java.io.BufferedReader;
import java.io.File;
import java.io.FileReader;
import java.io.IOException;
import java.io.RandomAccessFile;
import java.nio.ByteBuffer;
import java.nio.ByteOrder;
import java.nio.channels.FileChannel;
import java.util.Random;
import java.util.concurrent.ExecutorService;
import java.util.concurrent.Executors;

import mpicbg.models.AffineModel3D;
import net.imglib2.Cursor;
import net.imglib2.img.Img;
import net.imglib2.img.array.ArrayImg;
import net.imglib2.img.array.ArrayImgs;
import net.imglib2.img.basictypeaccess.array.FloatArray;
import net.imglib2.type.numeric.real.FloatType;
import net.preibisch.simulation.SimulateMultiViewDataset;
import net.preibisch.simulation.Tools;

public class DumpReference
{
	static float[] readFloats( final File f, final long n ) throws IOException
	{
		try ( RandomAccessFile raf = new RandomAccessFile( f, "r" ); FileChannel ch = raf.getChannel() )
		{
			if ( ch.size() != 4 * n )
				throw new IOException( f + ": expected " + 4 * n + " bytes, found " + ch.size() );
			final ByteBuffer bb = ByteBuffer.allocate( (int)( 4 * n ) ).order( ByteOrder.LITTLE_ENDIAN );
			while ( bb.hasRemaining() && ch.read( bb ) >= 0 ) {}
			bb.flip();
			final float[] out = new float[ (int)n ];
			bb.asFloatBuffer().get( out );
			return out;
		}
	}

	/** any Img in its own iteration order -- for the ArrayImgs the reference returns that is x fastest */
	static void writeImg( final File f, final Img< FloatType > img ) throws IOException
	{
		final ByteBuffer bb = ByteBuffer.allocate( (int)( 4 * img.size() ) ).order( ByteOrder.LITTLE_ENDIAN );
		final Cursor< FloatType > c = img.cursor();
		while ( c.hasNext() )
			bb.putFloat( c.next().get() );
		bb.flip();
		try ( RandomAccessFile raf = new RandomAccessFile( f, "rw" ); FileChannel ch = raf.getChannel() )
		{
			raf.setLength( 0 );
			while ( bb.hasRemaining() ) ch.write( bb );
		}
	}

	static void writeDoubles( final File f, final double... v ) throws IOException
	{
		final ByteBuffer bb = ByteBuffer.allocate( 8 * v.length ).order( ByteOrder.LITTLE_ENDIAN );
		for ( final double d : v ) bb.putDouble( d );
		bb.flip();
		try ( RandomAccessFile raf = new RandomAccessFile( f, "rw" ); FileChannel ch = raf.getChannel() )
		{
			raf.setLength( 0 );
			while ( bb.hasRemaining() ) ch.write( bb );
		}
	}

	static ArrayImg< FloatType, FloatArray > copyOf( final Img< FloatType > img )
	{
		final long[] dim = new long[ img.numDimensions() ];
		img.dimensions( dim );
		final float[] data = new float[ (int)img.size() ];
		final Cursor< FloatType > c = img.cursor();
		int i = 0;
		while ( c.hasNext() ) data[ i++ ] = c.next().get();
		return ArrayImgs.floats( data, dim );
	}

	public static void main( final String[] args ) throws Exception
	{
		if ( args.length != 2 )
		{
			System.err.println( "usage: DumpReference <dir with manifest.txt and inputs> <output dir>" );
			System.exit( 2 );
		}
		final File in = new File( args[ 0 ] ), out = new File( args[ 1 ] );
		out.mkdirs();
		final ExecutorService service = Executors.newFixedThreadPool( Runtime.getRuntime().availableProcessors() );
		try ( BufferedReader r = new BufferedReader( new FileReader( new File( in, "manifest.txt" ) ) ) )
		{
			String line;
			while ( ( line = r.readLine() ) != null )
			{
				line = line.trim();
				if ( line.isEmpty() || line.startsWith( "#" ) ) continue;
				final String[] t = line.split( "\\s+" );
				final String name = t[ 0 ];
				final long nx = Long.parseLong( t[ 1 ] ), ny = Long.parseLong( t[ 2 ] ), nz = Long.parseLong( t[ 3 ] );
				final long kx = Long.parseLong( t[ 4 ] ), ky = Long.parseLong( t[ 5 ] ), kz = Long.parseLong( t[ 6 ] );
				final int axis = Integer.parseInt( t[ 7 ] ), degrees = Integer.parseInt( t[ 8 ] );
				final double delta = Double.parseDouble( t[ 9 ] );
				final int inc = Integer.parseInt( t[ 10 ] );
				final float minValue = Float.parseFloat( t[ 11 ] ), target = Float.parseFloat( t[ 12 ] ), snr = Float.parseFloat( t[ 13 ] );
				final long seed = Long.parseLong( t[ 14 ] );

				final Img< FloatType > gt = ArrayImgs.floats( readFloats( new File( in, name + ".gt.raw" ), nx * ny * nz ), nx, ny, nz );
				final Img< FloatType > psf = ArrayImgs.floats( readFloats( new File( in, name + ".psf.raw" ), kx * ky * kz ), kx, ky, kz );

				final AffineModel3D m = SimulateMultiViewDataset.axisRotation( gt, axis, degrees );
				// the 3x4 matrix read off the model's action (applyInPlace is what the reference itself calls, :127): no assumption
				// about the layout of toArray / getMatrix
				final double[] o = m.apply( new double[] { 0, 0, 0 } );
				final double[] ex = m.apply( new double[] { 1, 0, 0 } ), ey = m.apply( new double[] { 0, 1, 0 } ), ez = m.apply( new double[] { 0, 0, 1 } );
				final double[] a = new double[ 12 ];
				for ( int row = 0; row < 3; ++row )
				{
					a[ 4 * row ] = ex[ row ] - o[ row ];
					a[ 4 * row + 1 ] = ey[ row ] - o[ row ];
					a[ 4 * row + 2 ] = ez[ row ] - o[ row ];
					a[ 4 * row + 3 ] = o[ row ];
				}
				writeDoubles( new File( out, name + ".affine.f64" ), a );
				final Img< FloatType > rot = SimulateMultiViewDataset.rotateAroundAxis( gt, axis, degrees );
				writeImg( new File( out, name + ".rot.raw" ), rot );
				final Img< FloatType > att = SimulateMultiViewDataset.attenuate3d( rot, delta );
				writeImg( new File( out, name + ".att.raw" ), att );
				final Img< FloatType > con = SimulateMultiViewDataset.convolve( att, psf, service );
				writeImg( new File( out, name + ".psf_norm.raw" ), psf );
				writeImg( new File( out, name + ".con_raw.raw" ), con );
				final double corr = Tools.adjustImage( con, minValue, target );
				writeImg( new File( out, name + ".con.raw" ), con );
				writeDoubles( new File( out, name + ".corr.f64" ), corr );
				final Img< FloatType > ext = SimulateMultiViewDataset.extractSlices( con, inc, -1.0f );
				writeImg( new File( out, name + ".ext.raw" ), ext );
				final Img< FloatType > iso = SimulateMultiViewDataset.makeIsotropic( ext, inc );
				writeImg( new File( out, name + ".iso.raw" ), iso );
				writeImg( new File( out, name + ".weight.raw" ), SimulateMultiViewDataset.computeWeightImage( iso, delta ) );
				final ArrayImg< FloatType, FloatArray > noisy = copyOf( ext );
				Tools.poissonProcess( noisy, snr, new Random( seed ) );
				writeImg( new File( out, name + ".poisson.raw" ), noisy );
				System.out.println( name + ": " + nx + "x" + ny + "x" + nz + " done, corr = " + corr );
			}
		}
		service.shutdown();
	}
}
