package net.preibisch.simulation.gpu;

import java.util.ArrayList;
import java.util.List;
import java.util.Random;

import net.imglib2.RandomAccessibleInterval;
import net.imglib2.img.Img;
import net.imglib2.type.numeric.real.FloatType;
import net.imglib2.view.Views;

/**
 * All GPUs of the node from ONE JVM: the view loop of SimulateMultiViewDataset.main (:567-613) sharded over the devices,
 * view v on device v % ndev (mvsim_group_*).  The ground truth goes host -> GPU 0 -> every GPU (RCCL scatter + all-gather
 * over xGMI, the only collective of the path), every view's acquisition comes back to the host.
 *
 * SOURCE ONLY in this repository (no JDK in the build image).
 */
public final class GpuGroup implements AutoCloseable
{
	private long handle;
	private long[] dim;

	public GpuGroup( final int ndev ) { handle = MvsimNative.groupCreate( ndev ); }

	public void setGroundTruth( final RandomAccessibleInterval< FloatType > groundTruth )
	{
		dim = Buffers.dims( groundTruth );
		try ( Buffers.Block gt = Buffers.toBlock( Views.zeroMin( groundTruth ) ) )
		{
			MvsimNative.groupBroadcastVolume( handle, gt.floats, dim );
		}
	}

	/** psfs.get( v ) is normalised in place; one nextLong() per view is drawn from rnd, in view order. */
	public List< Img< FloatType > > simulateViews( final List< Img< FloatType > > psfs, final int[] degrees, final double attenuation,
			final int lightsheetSpacing, final float poissonSNR, final Random rnd )
	{
		final int n = degrees.length;
		final long[] o = new long[] { dim[ 0 ], dim[ 1 ], ( dim[ 2 ] - 1 ) / lightsheetSpacing + 1 };
		final Buffers.Block[] p = new Buffers.Block[ n ], a = new Buffers.Block[ n ];
		final java.nio.FloatBuffer[] pf = new java.nio.FloatBuffer[ n ], af = new java.nio.FloatBuffer[ n ];
		final long[] seeds = new long[ n ];
		try
		{
			for ( int v = 0; v < n; ++v )
			{
				p[ v ] = Buffers.toBlock( psfs.get( v ) );
				a[ v ] = Buffers.direct( Buffers.size( o ) );
				pf[ v ] = p[ v ].floats;
				af[ v ] = a[ v ].floats;
				seeds[ v ] = rnd.nextLong();
			}
			MvsimNative.groupSimulateViews( handle, pf, Buffers.dims( psfs.get( 0 ) ), dim, degrees, attenuation, SimulateMultiViewDatasetGPU.minValue,
					SimulateMultiViewDatasetGPU.avgIntensity, lightsheetSpacing, poissonSNR, seeds, af );
			final List< Img< FloatType > > res = new ArrayList<>();
			for ( int v = 0; v < n; ++v )
			{
				Buffers.copyBack( p[ v ], psfs.get( v ) );
				res.add( Buffers.toImg( a[ v ], o ) );
			}
			return res;
		}
		finally
		{
			for ( int v = 0; v < n; ++v )
			{
				if ( p[ v ] != null ) p[ v ].close();
				if ( a[ v ] != null ) a[ v ].close();
			}
		}
	}

	@Override
	public void close()
	{
		if ( handle != 0 )
			MvsimNative.groupDestroy( handle );
		handle = 0;
	}
}
